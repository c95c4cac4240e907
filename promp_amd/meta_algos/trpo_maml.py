"""TRPOMAML (reference: meta_policy_search/meta_algos/trpo_maml.py:8-191): MAML with a TRPO outer step.
BASELINE.json config 5 (run_scripts/maml_run_mujoco.py).  exploration=True adds E-MAML's term (trpo_maml.py:137-144)."""
import numpy as np

from .. import _lib
from ..optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer, ExactDeviceHvp, FiniteDifferenceHvp
from ..utils import logger
from .base import MAMLAlgo


class _DeviceEvaluator(object):
    """The four graph evaluations the CG optimizer needs, each one pass of the device kernels over steps 0..K."""

    def __init__(self, algo):
        self.algo = algo
        self._memo = None
        self._gmemo = None       # (key, loss gradient): gradient() twice at one state is one device evaluation

    def _key(self):
        ctx = self.ctx
        return (id(ctx), ctx.state_version(), self.algo.inner_kind)

    def objectives_ride_on_gradient(self):
        """the loss gradient's pass computes the surrogate loss and the mean KL on its way (promp_meta_grad returns them): where that
        holds for the whole batch -- one process, or ranks under the library's communicator -- and nothing re-enters between the two
        (the E-MAML term rewrites step 0's advantages for a moment), asking for the gradient FIRST answers the optimizer's 'KL before' /
        'loss before' queries without a forward pass of their own"""
        return not self.algo.exploration and not self.algo.session.external()

    @property
    def ctx(self):
        return self.algo.session.ctx

    def _eta(self):
        return np.zeros(self.algo.num_inner_grad_steps, np.float32)

    def _whole_batch(self, g):
        """a gradient promp_meta_grad returned: the meta-batch's mean -- or, on ranks that exchange through the session's
        `collective` (no communicator in the context), this rank's share of it, summed here"""
        sess = self.algo.session
        return np.asarray(sess.collective(np.asarray(g, dtype=np.float64), 'sum'), dtype=np.float32) if sess.external() else g

    def _exploration(self, want_grad):
        """E-MAML term (trpo_maml.py:137-144): mean_i [ -mean(adj_avg_rewards_i) * mean_n log pi_theta(a0_n | s0_n) ], the
        log-likelihood of the INITIAL (step-0) actions under the pre-update parameters.  On the device that is the
        LOGLIK objective of slot 0 with unit advantages, evaluated at theta for every task."""
        a, ctx = self.algo, self.ctx
        saved = ctx.get_task_thetas()
        ctx.switch_to_pre_update()
        ctx.set_advantages(0, np.ones_like(a._explore_adv0))
        g, l, _ = ctx.eval_loss_grad(0, _lib.LOSS_LOGLIK, clip_log_std=True)      # l_i = -mean log pi ; g_i = d l_i / d theta
        ctx.set_advantages(0, a._explore_adv0)
        ctx.set_task_thetas(saved)
        c = a._explore_coeffs
        tot = np.concatenate([[np.dot(c, l.astype(np.float64))], c.dot(g.astype(np.float64)) if want_grad else []])
        n_global = a.session.M_global
        tot = a.session.allreduce(tot)                # task-sharded run: sum over the ranks
        return tot[0] / n_global, (tot[1:] / n_global if want_grad else None)

    def _objectives(self):
        """One forward evaluation of the meta-objective yields both the surrogate loss and the mean KL.  The optimizer asks for
        them one after the other at every parameter vector it visits (KL before / loss before, every line-search trial, loss
        after / KL after: conjugate_gradient_optimizer.py:227-236, trpo_maml.py:170-191); the second question is answered from
        the first one's pass as long as nothing the evaluation reads has been replaced (promp_state_version)."""
        ctx = self.ctx
        key = (id(ctx), ctx.state_version(), self.algo.inner_kind)
        if self._memo is None or self._memo[0] != key:
            r = self.algo.session.meta_eval(0.0, self._eta(), self.algo.inner_kind, _lib.OUTER_RATIO)
            self._memo = ((id(ctx), ctx.state_version(), self.algo.inner_kind), r)
        return self._memo[1]

    def loss(self):          # -mean_i mean(ratio * adv) at theta'_i   (trpo_maml.py:135,150)
        v = self._objectives()['loss']
        return v + self._exploration(False)[0] if self.algo.exploration else v

    def constraint_val(self):   # mean_i mean KL(old || pi_theta'_i)   (trpo_maml.py:133,147)
        return self._objectives()['outer_kl']

    def gradient(self):
        if self.objectives_ride_on_gradient():
            if self._gmemo is None or self._gmemo[0] != self._key():
                g, st = self.ctx.meta_grad(0.0, self._eta(), self.algo.inner_kind, _lib.OUTER_RATIO)
                key = self._key()
                self._gmemo = (key, g)
                if self._memo is None or self._memo[0] != key:
                    self._memo = (key, dict(loss=st['loss'], inner_kl=st['inner_kl'], outer_kl=st['outer_kl']))
            return self._gmemo[1]
        g = self._whole_batch(self.ctx.meta_grad(0.0, self._eta(), self.algo.inner_kind, _lib.OUTER_RATIO)[0])
        return (g + self._exploration(True)[1]).astype(np.float32) if self.algo.exploration else g

    def constraint_gradient(self):
        return self._whole_batch(self.ctx.meta_grad(0.0, self._eta(), self.algo.inner_kind, _lib.OUTER_KL)[0])

    def exact_hvp_available(self):
        """False on ranks that exchange through session.collective (no communicator in the context: promp_constraint_hvp refuses a
        shard's sums there); ExactDeviceHvp then falls back to the finite-difference products, whose gradients do cross it"""
        return not self.algo.session.external()

    def constraint_hvp(self, x, refresh_chain=True):   # exact (d2 constraint / d theta2) x on the device
        return self.ctx.constraint_hvp(np.asarray(x, dtype=np.float32), self.algo.inner_kind, refresh_chain)

    def cg_solve(self, b, cg_iters, reg_coeff, hvp_mode, eps=1e-5, residual_tol=1e-10):
        """ConjugateGradientOptimizer's solve with its products enqueued back to back on the device (promp_cg_solve):
        -> (direction, direction . (H + reg I) direction), or None where the library cannot run it (ranks that exchange through
        session.collective hold no communicator) and the optimizer loops over its own products on the host"""
        if self.algo.session.external():
            return None
        return self.ctx.cg_solve(np.asarray(b, dtype=np.float32), cg_iters, reg_coeff, eps, hvp_mode, residual_tol, self.algo.inner_kind)

    def get_theta(self):
        return self.ctx.get_theta()

    def set_theta(self, theta):
        self.ctx.set_theta(np.asarray(theta, dtype=np.float32))


class TRPOMAML(MAMLAlgo):
    """
    Args (trpo_maml.py:23-31): policy, name, step_size (trust region), inner_type in {'log_likelihood',
    'likelihood_ratio'}, exploration (E-MAML), inner_lr, meta_batch_size, num_inner_grad_steps,
    trainable_inner_step_size; plus hvp_approach in {'finite_difference' (the reference's, default), 'exact'}
    """

    def __init__(self, *args, name='trpo_maml', step_size=0.01, inner_type='likelihood_ratio', exploration=False,
                 hvp_approach='finite_difference', **kwargs):
        super(TRPOMAML, self).__init__(*args, **kwargs)
        assert hvp_approach in ('finite_difference', 'exact')
        assert inner_type in ['log_likelihood', 'likelihood_ratio', 'dice']
        if inner_type == 'dice':
            raise NotImplementedError          # as the reference (trpo_maml.py:63-64)
        self.step_size = step_size
        self.inner_type = inner_type
        self.inner_kind = _lib.INNER_LOGLIK if inner_type == 'log_likelihood' else _lib.INNER_RATIO
        self.name = name
        self.exploration = exploration
        self._optimization_keys = ['observations', 'actions', 'advantages', 'agent_infos']
        if exploration:      # trpo_maml.py:41-42
            self._optimization_keys.append('adj_avg_rewards')
        self.optimizer = ConjugateGradientOptimizer(
            hvp_approach=ExactDeviceHvp() if hvp_approach == 'exact' else FiniteDifferenceHvp())
        self.optimizer.build_graph(_DeviceEvaluator(self), step_size)

    def optimize_policy(self, all_samples_data, log=True):
        """trpo_maml.py:161-191"""
        assert len(all_samples_data) == self.num_inner_grad_steps + 1
        self._place_steps(all_samples_data)
        if self.exploration:
            last = all_samples_data[self.num_inner_grad_steps]
            self._explore_coeffs = np.array([np.mean(np.asarray(d['adj_avg_rewards'], dtype=np.float32)) for d in last], np.float64)
            self._explore_adv0 = np.concatenate([np.asarray(d['advantages'], dtype=np.float32) for d in all_samples_data[0]])
        if self.optimizer._ev.objectives_ride_on_gradient():
            self.optimizer.gradient()          # one pass: the gradient optimize() will ask for, and the two values asked for next
        logger.log('Computing KL before')
        mean_kl_before = self.optimizer.constraint_val()
        logger.log('Computing loss before')
        loss_before = self.optimizer.loss()
        logger.log('Optimizing')
        self.optimizer.optimize()
        logger.log('Computing loss after')
        loss_after = self.optimizer.loss()
        logger.log('Computing KL after')
        mean_kl = self.optimizer.constraint_val()
        if log:
            logger.logkv('MeanKLBefore', mean_kl_before)
            logger.logkv('MeanKL', mean_kl)
            logger.logkv('LossBefore', loss_before)
            logger.logkv('LossAfter', loss_after)
            logger.logkv('dLoss', loss_before - loss_after)
        self.last_stats = dict(mean_kl_before=mean_kl_before, mean_kl=mean_kl, loss_before=loss_before, loss_after=loss_after)
