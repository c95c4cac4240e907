"""MAMLAlgo (reference: meta_policy_search/meta_algos/base.py:88-313): inner adaptation step and the
bookkeeping that maps processed samples onto device-resident slabs."""
import numpy as np

from .. import _lib


class MetaAlgo(object):
    def __init__(self, policy):
        assert hasattr(policy, 'session') and hasattr(policy, 'update_task_parameters')
        self.policy = policy
        self._optimization_keys = None


class MAMLAlgo(MetaAlgo):
    """
    Args (meta_algos/base.py:100-113): policy, inner_lr=0.1, meta_batch_size=20, num_inner_grad_steps=1,
    trainable_inner_step_size=False
    """
    inner_kind = _lib.INNER_RATIO

    def __init__(self, policy, inner_lr=0.1, meta_batch_size=20, num_inner_grad_steps=1, trainable_inner_step_size=False):
        super(MAMLAlgo, self).__init__(policy)
        assert type(num_inner_grad_steps) and num_inner_grad_steps >= 0
        assert type(meta_batch_size) == int
        assert meta_batch_size == policy.meta_batch_size
        self.inner_lr = float(inner_lr)
        self.meta_batch_size = meta_batch_size
        self.num_inner_grad_steps = num_inner_grad_steps
        self.trainable_inner_step_size = trainable_inner_step_size   # never trained in the reference either (base.py:109,203)
        self.session = policy.session
        self.session.set_num_inner_steps(num_inner_grad_steps)
        # per-parameter step-size tensors initialised to inner_lr (base.py:303-313)
        self.step_sizes = np.full(self.session.theta.size, self.inner_lr, dtype=np.float32)
        self.session.set_step_sizes(self.step_sizes)
        self._adapt_count = 0

    def _slot_of(self, samples, default_slot):
        """device slot that holds `samples` (uploading them if they did not come from our sample processor)"""
        slot = self.session.resident_slot(samples)
        if slot is None:
            slot = default_slot % (self.num_inner_grad_steps + 1)
            self._upload_into(slot, samples)
        return slot

    def _place_steps(self, all_samples_data):
        """sampling step k into slot k for every k.  The context is sized for the largest step that has to be uploaded BEFORE
        the first upload: a context that grows between two uploads would drop the slabs uploaded so far."""
        missing = [sd for k, sd in enumerate(all_samples_data) if self.session.resident_slot(sd) != k]
        if missing:
            self.session.ensure(max(sum(len(d['advantages']) for d in sd) for sd in missing), self.meta_batch_size)
        for k, sd in enumerate(all_samples_data):
            if self.session.resident_slot(sd) != k:   # not resident, or resident in another slot (an extra process_samples call)
                self._upload_into(k, sd)

    def _upload_into(self, slot, samples):
        """host copy of `samples` -> `slot`; the samples remember where they now live, so the next call finds them resident
        instead of uploading again (and instead of displacing the step that owns the slot they used to point at)"""
        upload = self.session.upload_samples(slot, samples)
        for i, sd in enumerate(samples):
            if hasattr(sd, 'device_ref'):
                sd.device_ref = (self.session.serial, upload, slot, i)

    def _adapt(self, samples):
        """MAML inner step for each task; stores the adapted parameters in the policy (base.py:217-242)"""
        assert len(samples) == self.meta_batch_size
        if self.policy._pre_update_mode:
            self._adapt_count = 0
        slot = self._slot_of(samples, self._adapt_count)
        self.session.ensure()
        self.session.ctx.inner_adapt(slot, self.inner_kind)
        self._adapt_count += 1
        self.session.task_thetas = None
        self.session.param_version += 1
        self.policy._pre_update_mode = False
