"""DICE-MAML (reference: meta_policy_search/meta_algos/dice_maml.py:10-258).

Contract:
  * objective of BOTH the inner step and the meta-update:  -mean_{p,t}( magic_box(tau_{p,t}) * adjusted_reward_{p,t} * mask_{p,t} )
    with tau the cumulative log-likelihood along the path and magic_box(x) = exp(x - stop_gradient(x)) (:39-45, :245-258);
    the mean runs over the zero-padded [paths, max_path_length] array;
  * second-order MAML through the K inner steps (:84-152); one Adam step per optimize_policy (MAMLFirstOrderOptimizer default);
  * logs LossBefore / LossAfter = the meta-objective's value, which the magic box makes -mean(adjusted_reward * mask) of the last
    sampling step whatever the parameters.

On the device the gradient of that objective is the log-likelihood objective's with per-row suffix-sum weights, and its Hessian
adds a term that couples the time steps of a path; promp_set_dice_rewards / PROMP_INNER_DICE provide both (include/promp_hip.h).
"""
import numpy as np

from .. import _lib
from ..utils import logger
from .base import MAMLAlgo


class DICEMAML(MAMLAlgo):
    """Args (dice_maml.py:24-41): max_path_length, policy, name='dice_maml', learning_rate=1e-3, inner_lr, meta_batch_size,
    num_inner_grad_steps, trainable_inner_step_size"""
    inner_kind = _lib.INNER_DICE

    def __init__(self, max_path_length, *args, name='dice_maml', learning_rate=1e-3, **kwargs):
        super(DICEMAML, self).__init__(*args, **kwargs)
        self.max_path_length = max_path_length
        self.learning_rate = learning_rate
        self.name = name
        self._optimization_keys = ['observations', 'actions', 'adjusted_rewards', 'mask', 'agent_infos']
        self.last_stats = None

    def _flatten_dice(self, samples):
        """padded samples (list[M] of dicts with mask [P, T]) -> (flat slab of the valid rows, their DiCE rewards)"""
        assert len(samples) == self.session.M
        lens = []
        for sd in samples:
            m = np.asarray(sd['mask']) > 0.5
            n = m.sum(axis=1).astype(np.int64)
            assert all(m[p, :n[p]].all() for p in range(m.shape[0])), 'padding must follow the valid steps of a path'
            lens.append(n)
        sel = lambda key, sub=None: np.concatenate(
            [np.asarray(sd[key] if sub is None else sd[key][sub], dtype=np.float32)[p, :n[p]]
             for sd, n in zip(samples, lens) for p in range(len(n))])
        all_lens = np.concatenate(lens)
        fl = dict(task_path_offsets=np.concatenate([[0], np.cumsum([len(n) for n in lens])]).astype(np.int32),
                  path_row_offsets=np.concatenate([[0], np.cumsum(all_lens)]).astype(np.int32),
                  obs=sel('observations'), rew=np.zeros(int(all_lens.sum()), np.float32), act=sel('actions'),
                  old_mean=sel('agent_infos', 'mean'), old_log_std=sel('agent_infos', 'log_std'))
        # rows / (paths * max_path_length): the slab mean (1 / rows) becomes the reference's mean over the padded array
        rw = np.concatenate([np.concatenate([np.asarray(sd['adjusted_rewards'], dtype=np.float64)[p, :n[p]] for p in range(len(n))])
                             * (n.sum() / float(np.asarray(sd['mask']).size)) for sd, n in zip(samples, lens)])
        return fl, rw

    def _upload_dice(self, slot, samples, flat=None):
        fl, rw = flat if flat is not None else self._flatten_dice(samples)
        self.session.upload_flat(slot, fl)
        self.session.ctx.set_dice_rewards(slot, rw)

    def _slot_of(self, samples, default_slot):
        slot = self.session.resident_slot(samples)
        if slot is None:
            slot = default_slot % (self.num_inner_grad_steps + 1)
            self._upload_dice(slot, samples)
        return slot

    def _place_dice_steps(self, all_samples_data):
        """sampling step k into slot k with its DiCE rewards, for every k"""
        # steps that are not resident in their own slot are flattened first, so that the context is sized for the largest of
        # them before anything is uploaded (a context that grows in between would drop the slabs uploaded so far)
        todo = {k: self._flatten_dice(sd) for k, sd in enumerate(all_samples_data) if self.session.resident_slot(sd) != k}
        if todo:
            self.session.ensure(max(len(fl['rew']) for fl, _ in todo.values()),
                                max(len(fl['path_row_offsets']) - 1 for fl, _ in todo.values()))
            if any(self.session.resident_slot(sd) != k for k, sd in enumerate(all_samples_data) if k not in todo):
                todo = {k: self._flatten_dice(sd) for k, sd in enumerate(all_samples_data)}     # the context was re-created
            for k, flat in todo.items():
                self._upload_dice(k, all_samples_data[k], flat)

    def optimize_policy(self, all_samples_data, log=True):
        """MAML outer step (dice_maml.py:154-178)"""
        K = self.num_inner_grad_steps
        assert len(all_samples_data) == K + 1
        self._place_dice_steps(all_samples_data)
        ctx = self.session.ctx
        if log: logger.log('Optimizing')
        self.session.optimize(1, self.learning_rate, 0.0, np.zeros(K, np.float32), _lib.INNER_DICE, _lib.OUTER_LOGLIK)
        if log: logger.log('Computing statistics')
        # magic_box == 1 in value: the objective is minus the mean adjusted reward of the last step's valid entries
        # (mean over ALL tasks: a task-sharded run sums the ranks' per-task means; the gradient's all-reduce is inside optimize)
        per_task = [-np.mean(np.asarray(sd['adjusted_rewards']) * np.asarray(sd['mask'])) for sd in all_samples_data[K]]
        tot, cnt = self.session.allreduce([np.sum(per_task), len(per_task)])
        loss = float(tot / cnt)
        if log:
            logger.logkv('LossBefore', loss)
            logger.logkv('LossAfter', loss)
        self.last_stats = dict(loss_before=loss, loss_after=loss)
        self.session.param_version += 1
