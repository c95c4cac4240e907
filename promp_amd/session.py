"""Process-wide device session shared by the plugin objects.

The reference wires its objects through TensorFlow's global default session/graph
(tf.get_default_session(), meta_algos/base.py:227; policies/base.py:179).  Here the shared global is a
DeviceSession: one libpromp_hip context per process (= per GPU) that owns the trajectory slabs of sampling
steps 0..K and the policy / optimizer state.  MetaGaussianMLPPolicy creates it, MetaSampleProcessor and
ProMP find it through ``current()``.
"""
import numpy as np

from . import _lib

_current = None
_serial = 0


def current():
    return _current


class SamplesData(dict):
    """The dict process_samples returns (same 8 keys as the reference) plus a reference to the device-resident
    copy, so that _adapt / optimize_policy need no re-upload."""
    device_ref = None     # (session serial, upload serial, step slot, task index)


class DeviceSession:
    def __init__(self, meta_batch_size, obs_dim, action_dim, hidden_sizes, num_inner_grad_steps=1,
                 n_tasks_global=None, device_id=0, rank=0, world=1, hidden_act='tanh', output_act=None):
        global _current, _serial
        _serial += 1
        self.serial = _serial
        self.M, self.O, self.A = int(meta_batch_size), int(obs_dim), int(action_dim)
        self.hidden = tuple(int(h) for h in hidden_sizes)
        self.hidden_act = hidden_act
        self.output_act = output_act
        self.K = int(num_inner_grad_steps)
        self.M_global = int(n_tasks_global or meta_batch_size)
        self.device_id, self.rank, self.world = device_id, rank, world
        self.ctx = None
        self.capacity = (0, 0)
        self.theta = None
        self.step_sizes = None
        self.adam = None
        self.task_thetas = None
        self.step_cursor = 0
        self.upload_serial = [0] * (self.K + 1)
        self.param_version = 0        # bumped by whatever changes the tasks' parameters (host-side caches key on it)
        self.learn_std = True
        self.min_std = 1e-6
        self._upload_counter = 0      # never reset: a SamplesData from before a context re-creation can never match a later upload
        self._comm_ready = False      # the RCCL communicator is created once per session and moved across context re-creations
        # host-side statistics that span the whole meta-batch (reward moments, path statistics, E-MAML sums) cross the ranks
        # through allreduce(); `collective` replaces the library's RCCL communicator by any callable
        # (float64 array, 'sum' | 'max') -> array, e.g. a gloo all_reduce (tests, CPU-only launches)
        self.collective = None
        _current = self

    def allreduce(self, values, op='sum'):
        """element-wise sum / max over the ranks of a small float64 array (identity on one rank)"""
        values = np.ascontiguousarray(values, dtype=np.float64)
        if self.world == 1:
            return values
        if self.collective is not None:
            return np.asarray(self.collective(values.copy(), op), dtype=np.float64).reshape(values.shape)
        ctx = self.ensure()
        flat = values.reshape(-1)
        out = np.concatenate([ctx.allreduce_f64(flat[i:i + 64], op) for i in range(0, flat.size, 64)]) if flat.size else flat
        return out.reshape(values.shape)

    def external(self):
        """True when the meta-batch is sharded over ranks that exchange through `collective` (the context then holds no
        communicator and refuses promp_optimize / promp_constraint_hvp: its sums are one rank's)"""
        return self.world > 1 and self.collective is not None

    def _meta_eval_shares(self, ctx, clip_eps, eta, inner_kind, outer_kind):
        """one evaluation whose per-rank shares have been summed: (grad mean buffer applied?, stats dict)"""
        _, st = ctx.meta_grad(clip_eps, eta, inner_kind, outer_kind)       # local SUMS stay in the exchange buffer
        red = ctx.reduced_get()
        red = np.asarray(self.collective(red.astype(np.float64), 'sum'), dtype=np.float32)
        ctx.reduced_set(red)                                                # every rank now holds the meta-batch's sums
        K, n = self.K, float(self.M_global)
        eta = np.asarray(eta, dtype=np.float64)
        J, kls, okl = red[-(K + 2)] / n, red[-(K + 1):-1].astype(np.float64) / n, red[-1] / n
        return dict(loss=float(J + np.mean(eta * kls)), inner_kl=kls.astype(np.float32), outer_kl=float(okl))

    def optimize(self, num_epochs, lr, clip_eps, inner_kl_coeff, inner_kind=_lib.INNER_RATIO, outer_kind=_lib.OUTER_CLIP):
        """ProMP.optimize_policy's numerical core on this session: promp_optimize, or -- ranks that exchange through
        `collective` -- the same epochs with the [Theta + K + 2] buffer crossing the ranks on the host
        (promp_meta_grad -> promp_reduced_get -> collective -> promp_reduced_set -> promp_adam_step)."""
        ctx = self.ensure()
        if not self.external():
            return ctx.optimize(num_epochs, lr, clip_eps, inner_kl_coeff, inner_kind, outer_kind)
        eta = np.asarray(inner_kl_coeff, dtype=np.float32)
        loss_before = None
        for _ in range(int(num_epochs)):
            st = self._meta_eval_shares(ctx, clip_eps, eta, inner_kind, outer_kind)
            if loss_before is None:
                loss_before = st['loss']
            ctx.adam_step(lr)
        st = self._meta_eval_shares(ctx, clip_eps, eta, inner_kind, outer_kind)        # compute_stats
        return dict(loss_before=st['loss'] if loss_before is None else loss_before, loss_after=st['loss'],
                    inner_kl=st['inner_kl'], outer_kl=st['outer_kl'])

    def meta_eval(self, clip_eps, inner_kl_coeff, inner_kind=_lib.INNER_RATIO, outer_kind=_lib.OUTER_CLIP):
        r = self.optimize(0, 0.0, clip_eps, inner_kl_coeff, inner_kind, outer_kind)
        return dict(loss=r['loss_after'], inner_kl=r['inner_kl'], outer_kl=r['outer_kl'])

    def set_num_inner_steps(self, K):
        if K != self.K:
            self.K = int(K)
            self.upload_serial = [0] * (self.K + 1)
            self._drop()

    def _drop(self):
        if self.ctx is not None:
            self.pull_state()
            self.ctx.close()
            self.ctx = None
            self.capacity = (0, 0)
            self._comm_ready = False

    def pull_state(self):
        if self.ctx is not None:
            self.theta = self.ctx.get_theta()
            self.adam = self.ctx.get_adam_state()
            self.task_thetas = self.ctx.get_task_thetas()

    def ensure(self, rows=1, paths=1):
        """Context with room for `rows` rows / `paths` paths per sampling step (grown geometrically)."""
        if self.ctx is None or rows > self.capacity[0] or paths > self.capacity[1]:
            cap = (max(rows, int(1.25 * self.capacity[0])), max(paths, int(1.25 * self.capacity[1])))
            old = self.ctx
            if old is not None:
                self.pull_state()
            self.ctx = _lib.Context(self.M, self.O, self.A, self.hidden, self.K, max_rows=cap[0], max_paths=cap[1],
                                    n_tasks_global=self.M_global, device_id=self.device_id, hidden_act=self.hidden_act,
                                    output_act=self.output_act)
            if old is not None:
                # ranks with ragged batches regrow at different times: the communicator moves to the new context instead
                # of a fresh rendezvous (which would wait for peers that are not regrowing)
                if self._comm_ready:
                    self.ctx.comm_move_from(old)
                old.close()
            self.capacity = cap
            if self.theta is not None:
                self.ctx.set_theta(self.theta)
            if self.step_sizes is not None:
                self.ctx.set_step_sizes(self.step_sizes)
            self.ctx.set_learn_std(self.learn_std)
            self.ctx.set_min_std(self.min_std)
            if self.adam is not None:
                self.ctx.set_adam_state(*self.adam)
            if self.task_thetas is not None:
                self.ctx.set_task_thetas(self.task_thetas)
            if self.world > 1 and not self._comm_ready and self.collective is None:
                from . import comm
                self.ctx.comm_init(self.rank, self.world, comm.exchange_unique_id(self.rank, self.world, _lib.comm_unique_id))
                self._comm_ready = True
            self.upload_serial = [-1] * (self.K + 1)       # nothing is resident in the new context
        return self.ctx

    # ---- parameters ----
    def set_theta(self, theta):
        self.theta = np.ascontiguousarray(theta, dtype=np.float32)
        if self.ctx is not None:
            self.ctx.set_theta(self.theta)

    def get_theta(self):
        if self.ctx is not None:
            self.theta = self.ctx.get_theta()
        return self.theta

    def set_step_sizes(self, s):
        self.step_sizes = np.ascontiguousarray(s, dtype=np.float32)
        if self.ctx is not None:
            self.ctx.set_step_sizes(self.step_sizes)

    def next_slot(self):
        slot = self.step_cursor % (self.K + 1)
        self.step_cursor += 1
        return slot

    def upload_flat(self, slot, fl):
        ctx = self.ensure(len(fl['rew']), len(fl['path_row_offsets']) - 1)
        ctx.upload_step(slot, fl['task_path_offsets'], fl['path_row_offsets'], fl['obs'], fl['rew'], fl['act'],
                        fl['old_mean'], fl['old_log_std'])
        self._upload_counter += 1
        self.upload_serial[slot] = self._upload_counter
        return self.upload_serial[slot]

    def upload_samples(self, slot, samples_data_meta_batch):
        """Upload processed samples (list[M] of dicts) that did not come from this session's processor."""
        assert len(samples_data_meta_batch) == self.M
        n = [len(sd['advantages']) for sd in samples_data_meta_batch]
        cat = lambda f: np.concatenate([np.asarray(f(sd), dtype=np.float32) for sd in samples_data_meta_batch])
        fl = dict(task_path_offsets=np.arange(self.M + 1, dtype=np.int32),
                  path_row_offsets=np.concatenate([[0], np.cumsum(n)]).astype(np.int32),
                  obs=cat(lambda sd: sd['observations']).reshape(sum(n), -1), rew=np.zeros(sum(n), np.float32),
                  act=cat(lambda sd: sd['actions']).reshape(sum(n), -1),
                  old_mean=cat(lambda sd: sd['agent_infos']['mean']).reshape(sum(n), -1),
                  old_log_std=cat(lambda sd: sd['agent_infos']['log_std']).reshape(sum(n), -1))
        ser = self.upload_flat(slot, fl)
        self.ctx.set_advantages(slot, cat(lambda sd: sd['advantages']))
        return ser

    def resident_slot(self, samples_data_meta_batch):
        """Slot holding exactly these samples, or None."""
        refs = [getattr(sd, 'device_ref', None) for sd in samples_data_meta_batch]
        if any(r is None for r in refs) or len(refs) != self.M:
            return None
        ser, up, slot, _ = refs[0]
        if ser != self.serial or self.ctx is None or self.upload_serial[slot] != up:
            return None
        if any(r != (ser, up, slot, i) for i, r in enumerate(refs)):
            return None
        return slot
