"""The meta-policy-search iteration loop.

Drop-in for the reference's Trainer (meta_policy_search/meta_trainer.py:34-164): same constructor, `train()`, and the
same logger keys (Itr, n_timesteps, Time-*, ItrTime), so progress.csv files stay comparable.  One iteration is

    sample tasks -> for step 0..K: [rollouts -> process_samples -> (step < K) inner adaptation] -> outer update

`sess` is accepted for signature compatibility and ignored: there is no TensorFlow session, the device context lives
in the policy's DeviceSession.  Snapshots hold flat parameters and optimizer state instead of pickled graph objects.
"""
import time

from .utils import logger


class _Stopwatch(object):
    """accumulates wall-clock seconds under names"""

    def __init__(self):
        self.totals = {}

    def add(self, name, since):
        self.totals[name] = self.totals.get(name, 0.0) + (time.time() - since)

    def get(self, name):
        return self.totals.get(name, 0.0)


class Trainer(object):
    def __init__(self, algo, env, sampler, sample_processor, policy, n_itr, start_itr=0, num_inner_grad_steps=1, sess=None):
        self.algo, self.env, self.policy = algo, env, policy
        self.sampler, self.sample_processor = sampler, sample_processor
        self.baseline = sample_processor.baseline
        self.n_itr, self.start_itr = n_itr, start_itr
        self.num_inner_grad_steps = num_inner_grad_steps

    # ---- one sampling step of an iteration: rollouts, processing, diagnostics, inner adaptation ----
    def _sampling_step(self, step, watch):
        tag = 'Step_%d-' % step
        logger.log('** Step %d **' % step)
        logger.log('Obtaining samples...')
        since = time.time()
        paths = self.sampler.obtain_samples(log=True, log_prefix=tag)
        watch.add('sampling', since)
        logger.log('Processing samples...')
        since = time.time()
        samples_data = self.sample_processor.process_samples(paths, log='all', log_prefix=tag)
        watch.add('processing', since)
        self.log_diagnostics([path for task_paths in paths.values() for path in task_paths], prefix=tag)
        since = time.time()
        if step < self.num_inner_grad_steps:
            logger.log('Computing inner policy updates...')
            self.algo._adapt(samples_data)
        watch.add('inner', since)
        return samples_data

    def train(self):
        run_started = time.time()
        for itr in range(self.start_itr, self.n_itr):
            itr_started = time.time()
            watch = _Stopwatch()
            logger.log('\n ---------------- Iteration %d ----------------' % itr)
            logger.log('Sampling set of tasks/goals for this meta-batch...')
            self.sampler.update_tasks()
            self.policy.switch_to_pre_update()

            since = time.time()
            all_samples_data = None       # (meta_trainer.py:99 starts every iteration with empty lists: the previous iteration's
                                          #  samples are released BEFORE the new ones are made -- results handed out lazily that
                                          #  nobody read are then never downloaded)
            all_samples_data = [self._sampling_step(step, watch) for step in range(self.num_inner_grad_steps + 1)]
            watch.add('all_inner', since)

            logger.log('Optimizing policy...')
            outer_started = time.time()
            self.algo.optimize_policy(all_samples_data)
            outer_seconds = time.time() - outer_started

            for key, value in (('Itr', itr), ('n_timesteps', self.sampler.total_timesteps_sampled),
                               ('Time-OuterStep', outer_seconds), ('Time-TotalInner', watch.get('all_inner')),
                               ('Time-InnerStep', watch.get('inner')), ('Time-SampleProc', watch.get('processing')),
                               ('Time-Sampling', watch.get('sampling')), ('Time', time.time() - run_started),
                               ('ItrTime', time.time() - itr_started), ('Time-MAMLSteps', time.time() - outer_started)):
                logger.logkv(key, value)
            logger.log('Saving snapshot...')
            logger.save_itr_params(itr, self.get_itr_snapshot(itr))
            logger.log('Saved')
            logger.dumpkvs()
        logger.log('Training finished')

    def load_snapshot(self, snapshot):
        """Resume from what get_itr_snapshot wrote (a dict, or the path of a params.pkl / itr_<n>.pkl file): policy parameters,
        Adam moments and step count, inner KL coefficients, baseline coefficients; training continues at the next iteration."""
        if isinstance(snapshot, str):
            snapshot = logger.load_params(snapshot)
        self.policy.set_params(snapshot['policy_params'])
        self.policy.switch_to_pre_update()
        sess = self.policy.session
        if snapshot.get('adam_m') is not None:
            sess.adam = (snapshot['adam_m'], snapshot['adam_v'], int(snapshot['adam_t']))
            if sess.ctx is not None:
                sess.ctx.set_adam_state(*sess.adam)
        if snapshot.get('inner_kl_coeff') is not None and hasattr(self.algo, 'inner_kl_coeff'):
            self.algo.inner_kl_coeff = snapshot['inner_kl_coeff']
        if snapshot.get('baseline') is not None:
            self.baseline.set_params(snapshot['baseline'])
        self.start_itr = int(snapshot['itr']) + 1

    def get_itr_snapshot(self, itr):
        """what is needed to resume: flat policy parameters, Adam state, KL coefficients, baseline coefficients"""
        sess = self.policy.session
        ctx = sess.ctx
        adam_m, adam_v, adam_t = ctx.get_adam_state() if ctx is not None else (sess.adam or (None, None, 0))
        return dict(itr=itr, policy_params=self.policy.get_param_values(), adam_m=adam_m, adam_v=adam_v, adam_t=adam_t,
                    inner_kl_coeff=getattr(self.algo, 'inner_kl_coeff', None), baseline=self.baseline.get_param_values())

    def log_diagnostics(self, paths, prefix):
        for reporter in (self.env, self.policy, self.baseline):
            if hasattr(reporter, 'log_diagnostics'):
                reporter.log_diagnostics(paths, prefix)
