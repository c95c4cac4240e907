"""Trainer (reference: meta_policy_search/meta_trainer.py:8-164): the meta-policy-search iteration loop, with the
reference's timing / logging keys.  `sess` is accepted and ignored (there is no TensorFlow session)."""
import time

import numpy as np

from .utils import logger


class Trainer(object):
    def __init__(self, algo, env, sampler, sample_processor, policy, n_itr, start_itr=0, num_inner_grad_steps=1, sess=None):
        self.algo = algo
        self.env = env
        self.sampler = sampler
        self.sample_processor = sample_processor
        self.baseline = sample_processor.baseline
        self.policy = policy
        self.n_itr = n_itr
        self.start_itr = start_itr
        self.num_inner_grad_steps = num_inner_grad_steps

    def train(self):
        start_time = time.time()
        for itr in range(self.start_itr, self.n_itr):
            itr_start_time = time.time()
            logger.log('\n ---------------- Iteration %d ----------------' % itr)
            logger.log('Sampling set of tasks/goals for this meta-batch...')
            self.sampler.update_tasks()
            self.policy.switch_to_pre_update()
            all_samples_data, all_paths = [], []
            list_sampling_time, list_inner_step_time, list_proc_samples_time = [], [], []
            start_total_inner_time = time.time()
            for step in range(self.num_inner_grad_steps + 1):
                logger.log('** Step ' + str(step) + ' **')
                logger.log('Obtaining samples...')
                t0 = time.time()
                paths = self.sampler.obtain_samples(log=True, log_prefix='Step_%d-' % step)
                list_sampling_time.append(time.time() - t0)
                all_paths.append(paths)
                logger.log('Processing samples...')
                t0 = time.time()
                samples_data = self.sample_processor.process_samples(paths, log='all', log_prefix='Step_%d-' % step)
                all_samples_data.append(samples_data)
                list_proc_samples_time.append(time.time() - t0)
                self.log_diagnostics(sum(list(paths.values()), []), prefix='Step_%d-' % step)
                t0 = time.time()
                if step < self.num_inner_grad_steps:
                    logger.log('Computing inner policy updates...')
                    self.algo._adapt(samples_data)
                list_inner_step_time.append(time.time() - t0)
            total_inner_time = time.time() - start_total_inner_time
            time_maml_opt_start = time.time()
            logger.log('Optimizing policy...')
            time_outer_step_start = time.time()
            self.algo.optimize_policy(all_samples_data)
            logger.logkv('Itr', itr)
            logger.logkv('n_timesteps', self.sampler.total_timesteps_sampled)
            logger.logkv('Time-OuterStep', time.time() - time_outer_step_start)
            logger.logkv('Time-TotalInner', total_inner_time)
            logger.logkv('Time-InnerStep', np.sum(list_inner_step_time))
            logger.logkv('Time-SampleProc', np.sum(list_proc_samples_time))
            logger.logkv('Time-Sampling', np.sum(list_sampling_time))
            logger.logkv('Time', time.time() - start_time)
            logger.logkv('ItrTime', time.time() - itr_start_time)
            logger.logkv('Time-MAMLSteps', time.time() - time_maml_opt_start)
            logger.log('Saving snapshot...')
            logger.save_itr_params(itr, self.get_itr_snapshot(itr))
            logger.log('Saved')
            logger.dumpkvs()
        logger.log('Training finished')

    def get_itr_snapshot(self, itr):
        """flat parameters + optimizer state instead of pickled TF objects"""
        s = self.policy.session
        m, v, t = s.ctx.get_adam_state() if s.ctx is not None else (None, None, 0)
        return dict(itr=itr, policy_params=self.policy.get_param_values(), adam_m=m, adam_v=v, adam_t=t,
                    inner_kl_coeff=getattr(self.algo, 'inner_kl_coeff', None),
                    baseline=self.baseline.get_param_values())

    def log_diagnostics(self, paths, prefix):
        if hasattr(self.env, 'log_diagnostics'):
            self.env.log_diagnostics(paths, prefix)
        self.policy.log_diagnostics(paths, prefix)
        self.baseline.log_diagnostics(paths, prefix)
