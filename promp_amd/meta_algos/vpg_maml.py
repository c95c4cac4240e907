"""VPG-MAML: MAML with a vanilla policy-gradient outer objective and one Adam step per iteration.

Contract (reference: meta_policy_search/meta_algos/vpg_maml.py:9-175, optimizers/maml_first_order_optimizer.py:22-115):
  * inner objective  'likelihood_ratio': -mean(ratio * adv)  or  'log_likelihood': -mean(log pi * adv);
  * meta-objective   mean_i [ -mean_n( log pi_theta'_i(a|s) * adv ) ]  on the last sampling step's data, differentiated
    through the adaptation (second-order MAML, like every MAMLAlgo);
  * exploration=True adds E-MAML's pre-update term  -mean(adj_avg_rewards_i) * mean_n log pi_theta(a0|s0);
  * optimize_policy: ONE Adam step (max_epochs = 1); logs LossBefore (the loss that step saw) and LossAfter.

On the device this is the ProMP pipeline with other objective kinds (inner kind as configured, outer kind LOGLIK, no KL
penalty): promp_optimize with a single epoch.  With the exploration term the gradient is assembled in two pieces -- the
meta-gradient on the device, the log-likelihood gradient of the step-0 actions at theta -- summed in the exchange buffer
(promp_reduced_get / _set) before promp_adam_step.
"""
import numpy as np

from .. import _lib
from ..utils import logger
from .base import MAMLAlgo


class VPGMAML(MAMLAlgo):
    def __init__(self, *args, name='vpg_maml', learning_rate=1e-3, inner_type='likelihood_ratio', exploration=False, **kwargs):
        super(VPGMAML, self).__init__(*args, **kwargs)
        assert inner_type in ('log_likelihood', 'likelihood_ratio')
        self.name, self.learning_rate = name, learning_rate
        self.inner_type, self.exploration = inner_type, exploration
        self.inner_kind = _lib.INNER_LOGLIK if inner_type == 'log_likelihood' else _lib.INNER_RATIO
        self._optimization_keys = ['observations', 'actions', 'advantages', 'agent_infos'] + (['adj_avg_rewards'] if exploration else [])
        self.last_stats = None

    def _exploration_term(self, ctx, coeffs, adv0, want_grad):
        """sum_i coeffs_i * ( -mean log pi_theta(a0|s0) ) and its gradient wrt theta: the LOGLIK objective of slot 0 with unit
        advantages at the pre-update parameters"""
        saved = ctx.get_task_thetas()
        ctx.switch_to_pre_update()
        ctx.set_advantages(0, np.ones_like(adv0))
        g, l, _ = ctx.eval_loss_grad(0, _lib.LOSS_LOGLIK, clip_log_std=True)
        ctx.set_advantages(0, adv0)
        ctx.set_task_thetas(saved)
        return float(np.dot(coeffs, l.astype(np.float64))), (coeffs.dot(g.astype(np.float64)) if want_grad else None)

    def optimize_policy(self, all_samples_data, log=True):
        K = self.num_inner_grad_steps
        assert len(all_samples_data) == K + 1
        self._place_steps(all_samples_data)
        ctx, sess = self.session.ctx, self.session
        eta = np.zeros(K, np.float32)
        if log: logger.log('Optimizing')
        if not self.exploration:
            res = sess.optimize(1, self.learning_rate, 0.0, eta, self.inner_kind, _lib.OUTER_LOGLIK)
            loss_before, loss_after = res['loss_before'], res['loss_after']
        else:
            coeffs = np.array([np.mean(np.asarray(d['adj_avg_rewards'], dtype=np.float32)) for d in all_samples_data[K]], np.float64)
            adv0 = np.concatenate([np.asarray(d['advantages'], dtype=np.float32) for d in all_samples_data[0]])
            n_global = sess.M_global

            def total(want_grad):
                v, g = self._exploration_term(ctx, coeffs, adv0, want_grad)
                tot = np.concatenate([[v], g if want_grad else []])
                tot = sess.allreduce(tot)                 # task-sharded run: sum over the ranks
                return tot[0] / n_global, (tot[1:] if want_grad else None)
            _, st = ctx.meta_grad(0.0, eta, self.inner_kind, _lib.OUTER_LOGLIK)         # sums over the local tasks stay in the buffer
            x_val, x_grad_sum = total(True)
            loss_before = st['loss'] + x_val
            red = ctx.reduced_get()
            red[:x_grad_sum.size] += x_grad_sum.astype(np.float32)          # both are sums over ALL tasks (all-reduced)
            ctx.reduced_set(red)
            ctx.adam_step(self.learning_rate)
            if log: logger.log('Computing statistics')
            loss_after = sess.meta_eval(0.0, eta, self.inner_kind, _lib.OUTER_LOGLIK)['loss'] + total(False)[0]
        if log:
            logger.logkv('LossBefore', loss_before)
            logger.logkv('LossAfter', loss_after)
        self.last_stats = dict(loss_before=loss_before, loss_after=loss_after)
        self.session.param_version += 1
