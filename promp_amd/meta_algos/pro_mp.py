"""ProMP (reference: meta_policy_search/meta_algos/pro_mp.py:9-214 + optimizers/maml_first_order_optimizer.py)."""
import numpy as np

from .. import _lib
from ..optimizers.maml_first_order_optimizer import MAMLPPOOptimizer
from ..utils import logger
from .base import MAMLAlgo


class ProMP(MAMLAlgo):
    """
    Args (pro_mp.py:30-57): policy, name, learning_rate, num_ppo_steps, num_minibatches (unused, as in the
    reference: maml_first_order_optimizer.py:41,97-99), clip_eps, target_inner_step, init_inner_kl_penalty,
    adaptive_inner_kl_penalty, anneal_factor (unused in the reference too: clip_eps is fed un-annealed,
    pro_mp.py:182), inner_lr, meta_batch_size, num_inner_grad_steps, trainable_inner_step_size
    """
    inner_kind = _lib.INNER_RATIO
    outer_kind = _lib.OUTER_CLIP

    def __init__(self, *args, name='ppo_maml', learning_rate=1e-3, num_ppo_steps=5, num_minibatches=1, clip_eps=0.2,
                 target_inner_step=0.01, init_inner_kl_penalty=1e-2, adaptive_inner_kl_penalty=True, anneal_factor=1.0,
                 **kwargs):
        super(ProMP, self).__init__(*args, **kwargs)
        self.learning_rate = learning_rate
        self.num_ppo_steps = num_ppo_steps
        self.num_minibatches = num_minibatches
        self.clip_eps = clip_eps
        self.target_inner_step = target_inner_step
        self.adaptive_inner_kl_penalty = adaptive_inner_kl_penalty
        self.inner_kl_coeff = init_inner_kl_penalty * np.ones(self.num_inner_grad_steps)
        self.anneal_coeff = 1
        self.anneal_factor = anneal_factor
        self._optimization_keys = ['observations', 'actions', 'advantages', 'agent_infos']
        self.name = name
        # pro_mp.py:59-62: the optimiser object a caller may reach for (algo.optimizer.loss / optimize / compute_stats)
        self.optimizer = MAMLPPOOptimizer(learning_rate=learning_rate, max_epochs=num_ppo_steps, num_minibatches=num_minibatches)
        self.optimizer.build_graph(self)

    def optimize_policy(self, all_samples_data, log=True):
        """MAML outer step: E Adam epochs on the meta-objective, then stats (pro_mp.py:165-199)"""
        assert len(all_samples_data) == self.num_inner_grad_steps + 1
        self._place_steps(all_samples_data)           # sampling step k must sit in slot k
        if log: logger.log('Optimizing')
        # (the epochs and the statistics pass behind them are ONE device call: compute_stats answers from what optimize got back)
        feed = dict(clip_eps=self.clip_eps, inner_kl_coeff=self.inner_kl_coeff)
        self.optimizer._learning_rate, self.optimizer._max_epochs = float(self.learning_rate), int(self.num_ppo_steps)
        loss_before = self.optimizer.optimize(feed)
        if log: logger.log('Computing statistics')
        loss_after, inner_kls, outer_kl = self.optimizer.compute_stats(feed)
        res = self.optimizer.last_result
        if self.adaptive_inner_kl_penalty:
            if log: logger.log('Updating inner KL loss coefficients')
            self.inner_kl_coeff = self.adapt_kl_coeff(self.inner_kl_coeff, inner_kls, self.target_inner_step)
        if log:
            logger.logkv('LossBefore', loss_before)
            logger.logkv('LossAfter', loss_after)
            logger.logkv('KLInner', np.mean(inner_kls))
            logger.logkv('KLCoeffInner', np.mean(self.inner_kl_coeff))
        self.last_stats = res

    def adapt_kl_coeff(self, kl_coeff, kl_values, kl_target):
        """per inner step: one coefficient per KL value (arrays), or a single pair (scalars)"""
        if np.ndim(kl_values) == 0:
            return _adapt_kl_coeff(kl_coeff, kl_values, kl_target)
        assert len(kl_coeff) == len(kl_values)
        return np.array([_adapt_kl_coeff(c, kl, kl_target) for c, kl in zip(kl_coeff, kl_values)])


def _adapt_kl_coeff(kl_coeff, kl, kl_target):
    """Penalty schedule of the inner KL term (pro_mp.py:201-214): a step whose KL stayed clearly inside the target
    (below target / 1.5) gets half the penalty next time, one clearly outside (above 1.5 target) twice the penalty."""
    too_small, too_large = kl < kl_target / 1.5, kl > kl_target * 1.5
    return kl_coeff / 2 if too_small else kl_coeff * 2 if too_large else kl_coeff
