"""VPG_DICEMAML (reference: meta_policy_search/meta_algos/vpg_dice_maml.py:7-127): DiCE inner steps under a vanilla-policy-gradient
outer objective.

    inner (per task, steps 0..K-1):  the DiCE objective of DICEMAML (dice_maml.py:39-45, 245-258)
    outer:                           -mean(log pi_theta'(a|s) * advantage * mask) over the last step's padded samples
                                     (vpg_dice_maml.py:98-104), mean over the tasks; one Adam step (maml_first_order_optimizer.py)

The samples come from DiceMetaSampleProcessor with a return_baseline (which adds the GAE 'advantages' the outer objective
reads).  On the device this is promp_optimize with PROMP_INNER_DICE / PROMP_OUTER_LOGLIK -- the kernels of DICE-MAML -- with the
last step's gradient weights replaced by the advantages instead of the DiCE suffix sums.
"""
import numpy as np

from .. import _lib
from ..utils import logger
from .dice_maml import DICEMAML


class VPG_DICEMAML(DICEMAML):
    """Args (vpg_dice_maml.py:21-28): max_path_length, policy, name='vpg_dice_maml', learning_rate=1e-3, inner_lr,
    meta_batch_size, num_inner_grad_steps, trainable_inner_step_size"""

    def __init__(self, max_path_length, *args, name='vpg_dice_maml', **kwargs):
        super(VPG_DICEMAML, self).__init__(max_path_length, *args, name=name, **kwargs)
        self._optimization_keys = ['observations', 'actions', 'advantages', 'adjusted_rewards', 'mask', 'agent_infos']

    def optimize_policy(self, all_samples_data, log=True):
        """vpg_dice_maml.py:35-113 (graph) + dice_maml.py:154-178 (outer step)"""
        K = self.num_inner_grad_steps
        assert len(all_samples_data) == K + 1
        assert all('advantages' in sd for sd in all_samples_data[K]), \
            "the last step's samples carry no 'advantages': build the DiCE sample processor with a return_baseline"
        self._place_dice_steps(all_samples_data)
        ctx = self.session.ctx
        # the last step's weights: advantage * mask of the valid entries; rows / (paths * max_path_length) turns the slab mean
        # (1 / rows) into the reference's mean over the padded array
        adv = []
        for sd in all_samples_data[K]:
            m = np.asarray(sd['mask']) > 0.5
            n = m.sum(axis=1).astype(np.int64)
            a = np.asarray(sd['advantages'], dtype=np.float64)
            adv.append(np.concatenate([a[p, :n[p]] for p in range(len(n))]) * (n.sum() / float(m.size)))
        ctx.set_advantages(K, np.concatenate(adv).astype(np.float32))
        if log: logger.log('Optimizing')
        res = self.session.optimize(1, self.learning_rate, 0.0, np.zeros(K, np.float32), _lib.INNER_DICE, _lib.OUTER_LOGLIK)
        # slot K's weights are the GAE advantages now, not the DiCE suffix sums promp_set_dice_rewards wrote: the samples are no
        # longer "resident" as DiCE data (a later DiCE use of the same samples re-uploads instead of differentiating the wrong
        # objective)
        self.session.upload_serial[K] = -1
        if log: logger.log('Computing statistics')
        if log:
            logger.logkv('LossBefore', res['loss_before'])
            logger.logkv('LossAfter', res['loss_after'])
        self.last_stats = dict(loss_before=res['loss_before'], loss_after=res['loss_after'])
        self.session.param_version += 1
