"""SampleProcessor (reference: meta_policy_search/samplers/base.py:33-173), single-task flavour.
The arithmetic runs on the device; see MetaSampleProcessor for the meta-batch version the Trainer uses."""
from collections import OrderedDict

import numpy as np

from .. import _lib, session as session_mod
from ..utils import logger


def _concat_tensor_dict_list(dict_list):
    """utils/utils.py:104-122"""
    if not dict_list or not dict_list[0]:
        return {}
    out = {}
    for k in dict_list[0].keys():
        ex = dict_list[0][k]
        out[k] = _concat_tensor_dict_list([d[k] for d in dict_list]) if isinstance(ex, dict) \
            else np.concatenate([d[k] for d in dict_list])
    return out


def _shared_storage(fl, first):
    """True when per-task slices of the flat arrays `fl` equal what concatenating the path dicts' arrays would give, dtype and
    shape included (flatten_paths casts to float32 and reshapes to [n, dim]; anything it changed is concatenated the old way)"""
    if fl.get('obs') is None or fl.get('act') is None or fl.get('old_mean') is None or fl.get('rew') is None:
        return False
    if 'agent_infos' not in first or not first['agent_infos'] or 'mean' not in first['agent_infos']:
        return False
    same = lambda a, ref: isinstance(a, np.ndarray) and a.dtype == ref.dtype and a.ndim == ref.ndim and a.shape[1:] == ref.shape[1:]
    return same(first['observations'], fl['obs']) and same(first['actions'], fl['act']) and same(first['rewards'], fl['rew']) \
        and same(first['agent_infos']['mean'], fl['old_mean']) and same(first['agent_infos']['log_std'], fl['old_mean'])


class SampleProcessor(object):
    """
    - fits a reward baseline, - performs GAE, - stacks the path data, - logs path statistics

    Args (samplers/base.py:48-65): baseline, discount=0.99, gae_lambda=1, normalize_adv=False, positive_adv=False
    """

    # OPT-IN (default False = the reference's behaviour: plain ndarrays, and every path dict receives 'returns' / 'advantages' inside
    # process_samples, samplers/base.py:104,159).  True: a batch that is already resident on the device (DevicePaths from
    # DeviceSlabSampler / DevicePointEnvSampler) gets its per-row results back LAZILY -- 'returns' / 'advantages' of the samples
    # data and of the path dicts are _lib.LazyRows that cross PCIe on first use (np.asarray, arithmetic, indexing ...), which the
    # training loop never does (_adapt / optimize_policy work on the resident copy), and the path-dict side effect is applied when
    # the DevicePaths container is next looked at.  A caller that kept references to the path lists / dicts from BEFORE the call, or
    # tests isinstance(x, np.ndarray), must leave this off.
    lazy_host_arrays = False

    def __init__(self, baseline, discount=0.99, gae_lambda=1, normalize_adv=False, positive_adv=False):
        assert 0 <= discount <= 1.0, 'discount factor must be in [0,1]'
        assert 0 <= gae_lambda <= 1.0, 'gae_lambda must be in [0,1]'
        assert hasattr(baseline, 'fit') and hasattr(baseline, 'predict')
        self.baseline = baseline
        self.discount = discount
        self.gae_lambda = gae_lambda
        self.normalize_adv = normalize_adv
        self.positive_adv = positive_adv
        self._private_session = None

    # -- device plumbing -------------------------------------------------------------------------
    def _session_for(self, M, O, A):
        s = session_mod.current()
        if s is not None and (s.M, s.O, s.A) == (M, O, A):
            return s
        p = self._private_session
        if p is None or (p.M, p.O, p.A) != (M, O, A):
            keep = session_mod.current()
            p = session_mod.DeviceSession(M, O, A, (32, 32), 1)   # private: sample processing only
            session_mod._current = keep
            self._private_session = p
        return p

    def _stat_session(self):
        """the session whose ranks share this processor's meta-batch (statistics over all tasks cross it)"""
        return session_mod.current() or self._private_session or _OneRank

    def _process_on_device(self, paths_meta_batch):
        """-> (list[M] of SamplesData, per-path float64 stats) ; mutates the path dicts like the reference.

        Host work is kept off the per-path / per-task level where the API allows it: the flat arrays that went to the device
        (or came from it) are the storage of every task's observations / actions / rewards / agent_infos -- the per-task
        entries are VIEWS of them, not fresh concatenations (the reference concatenates, samplers/base.py:165-173; the values
        are the same) -- and the per-path 'returns' / 'advantages' the reference leaves in the path dicts (base.py:104,159) are
        views of the two downloaded float64 arrays."""
        M = len(paths_meta_batch)
        if hasattr(paths_meta_batch, 'settle'):
            paths_meta_batch._pending = None      # an earlier call's side effect nobody looked at: this call's replaces it
        sess, ref = session_mod.current(), getattr(paths_meta_batch, 'device_ref', None)
        # (the path lists as they are: looking at them through a DevicePaths' public methods would settle a pending side effect)
        path_lists = list(paths_meta_batch.raw_values() if hasattr(paths_meta_batch, 'raw_values') else paths_meta_batch.values())
        resident = ref is not None and sess is not None and sess.ctx is not None and ref[0] == sess.serial \
            and sess.upload_serial[ref[2]] == ref[1]
        kind = getattr(self.baseline, 'kind', _lib.BASELINE_ZERO)
        opts = dict(discount=self.discount, gae_lambda=self.gae_lambda, normalize_adv=self.normalize_adv,
                    positive_adv=self.positive_adv, baseline_kind=kind, reg_coeff=getattr(self.baseline, '_reg_coeff', 1e-5))
        if resident:
            # a device rollout (samplers/device_point_sampler.py): the slab is already resident, nothing to upload
            fl, upload, slot = paths_meta_batch.flat, ref[1], ref[2]
            sess.ctx.process_samples(slot, **opts)
        else:
            lib = _lib.get_library()
            first = path_lists[0][0]
            A = int(np.asarray(first['actions']).reshape(len(first['rewards']), -1).shape[1]) if 'actions' in first else 1
            # samplers.meta_sampler.HostPaths: the sampler built the flat arrays while it collected, and they are what goes to the
            # device as long as every path dict still holds the sampler's own arrays.  Checking that is 800 paths x 6 identities:
            # the copies and the kernels are enqueued FIRST (page-locked sources: promp_upload_step returns at once) and the check
            # runs while they are under way; a batch somebody edited since (rare) is flattened the general way and goes again.
            verify, fl = getattr(paths_meta_batch, 'flat_if_intact', None), getattr(paths_meta_batch, 'flat', None)
            optimistic = verify is not None and fl is not None and len(fl['task_path_offsets']) == M + 1 \
                and int(fl['task_path_offsets'][-1]) == sum(len(pl) for pl in path_lists)
            if not optimistic:
                fl = _lib.flatten_paths(paths_meta_batch, lib)
            sess = self._session_for(M, fl['obs'].shape[1], A)
            slot = sess.next_slot()
            upload = sess.upload_flat(slot, fl)
            sess.ctx.process_samples(slot, **opts)
            if optimistic and verify() is None:
                fl = _lib.flatten_paths(paths_meta_batch, lib)
                upload = sess.upload_flat(slot, fl)
                sess.ctx.process_samples(slot, **opts)
        ctx = sess.ctx
        lazy = bool(resident and self.lazy_host_arrays and hasattr(paths_meta_batch, 'settle'))
        # coefficients and per-path sums always come back (small); the per-row arrays now or on first use.  Their destination
        # arrays exist from here on; their CONTENTS arrive with the blocking fetch at the end of this function -- everything in
        # between (two views per path, seven per task) is host work done while the device is still busy with this call's upload
        # and kernels.
        out = ctx.alloc_processed(slot, kind, want_returns32=False, want_advantages=not lazy)
        pro, tpo = fl['path_row_offsets'], fl['task_path_offsets']
        n_rows = int(pro[-1])
        if lazy:
            res = ctx.lazy_results(slot)
            rows = lambda field, a, b: _lib.LazyRows(res, field, a, b)
            ret64, raw_adv64, adv32 = rows('returns', 0, n_rows), rows('raw_advantages', 0, n_rows), rows('advantages', 0, n_rows)
        else:
            ret64, raw_adv64 = ctx.alloc_raw(slot)
            adv32 = out['advantages']
            rows = lambda field, a, b: dict(returns=ret64, raw_advantages=raw_adv64, advantages=adv32)[field][a:b]
        # side effect of samplers/base.py:104,159: two views and two dict stores per path (plain slices on Python ints: np.split
        # costs ten times as much per piece) -- now, or when somebody looks at the paths of a resident batch again
        flat_paths = [p for plist in path_lists for p in plist]
        ends = pro.tolist()

        def side_effect():
            a = ends[0]
            if lazy:
                for p, b in zip(flat_paths, ends[1:]):
                    p['returns'] = rows('returns', a, b)
                    p['advantages'] = rows('raw_advantages', a, b)
                    a = b
            else:
                for p, b in zip(flat_paths, ends[1:]):
                    p['returns'] = ret64[a:b]
                    p['advantages'] = raw_adv64[a:b]
                    a = b
        if lazy:
            paths_meta_batch._pending = side_effect
        else:
            side_effect()
        first = flat_paths[0]
        shared = _shared_storage(fl, first)       # the flat arrays hold exactly what the path dicts hold (dtype and all)
        row0 = pro[tpo].tolist()
        result = []
        for i, plist in enumerate(path_lists):
            r0, r1 = row0[i], row0[i + 1]
            if shared:
                ls = fl['old_log_std']
                infos = dict(mean=fl['old_mean'][r0:r1],
                             log_std=ls[r0:r1] if len(ls) == len(fl['obs']) else np.broadcast_to(ls[i], (r1 - r0, ls.shape[1])))
                extra = [k for k in first['agent_infos'] if k not in ('mean', 'log_std')]
                if extra:
                    more = _concat_tensor_dict_list([{k: p['agent_infos'][k] for k in extra} for p in plist])
                    infos.update(more)
                obs, act, rew = fl['obs'][r0:r1], fl['act'][r0:r1], fl['rew'][r0:r1]
            else:
                obs = np.concatenate([p['observations'] for p in plist])
                act = np.concatenate([p['actions'] for p in plist])
                rew = np.concatenate([p['rewards'] for p in plist])
                infos = _concat_tensor_dict_list([p.get('agent_infos', {}) for p in plist])
            sd = session_mod.SamplesData(
                observations=obs, actions=act, rewards=rew, returns=rows('returns', r0, r1), advantages=rows('advantages', r0, r1),
                env_infos=_concat_tensor_dict_list([p.get('env_infos', {}) for p in plist]) if first.get('env_infos') else {},
                agent_infos=infos,
            )
            sd.device_ref = (sess.serial, upload, slot, i)
            result.append(sd)
        ctx.fetch_processed(slot, out)
        if lazy:
            out['advantages'] = adv32
        else:
            ctx.fetch_raw(slot, ret64, raw_adv64)
        if kind != _lib.BASELINE_ZERO:
            self.baseline._coeffs = out['coeffs'][-1].copy()      # the shared baseline ends on the last task's fit
        return result, out

    # -- reference API ---------------------------------------------------------------------------
    def process_samples(self, paths, log=False, log_prefix=''):
        """single task: list of paths -> dict with 7 keys (samplers/base.py:67-95)"""
        assert type(paths) == list, 'paths must be a list'
        assert paths[0].keys() >= {'observations', 'actions', 'rewards'}
        assert self.baseline, 'baseline must be specified - use self.build_sample_processor(baseline_obj)'
        result, out = self._process_on_device(OrderedDict([(0, paths)]))
        self._log_path_stats(out, log=log, log_prefix='')      # the reference drops log_prefix here (base.py:92)
        sd = dict(result[0])
        assert sd.keys() >= {'observations', 'actions', 'rewards', 'advantages', 'returns'}
        return sd

    def _log_path_stats(self, out, log=False, log_prefix=''):
        """samplers/base.py:135-149, from the per-path sums the device produced"""
        _log_return_stats(self._stat_session(), out['path_undiscounted'], out['path_returns0'], log, log_prefix)


class _OneRank(object):
    """stands in for a session when none exists: nothing to reduce over"""
    world = 1

    @staticmethod
    def allreduce(values, op='sum'):
        return np.asarray(values, dtype=np.float64)


def _log_return_stats(sess, und, disc, log, log_prefix):
    """The six path statistics of samplers/base.py:135-149 over ALL paths of the meta-batch.  On one rank they are the NumPy
    reductions the reference calls; a task-sharded run combines the ranks' counts, sums and extrema."""
    if not (log == 'reward' or log == 'all' or log is True):
        return
    und, disc = np.asarray(und, dtype=np.float64), np.asarray(disc, dtype=np.float64)
    if sess.world == 1:
        n, mean, mean_disc, std, hi, lo = len(und), np.mean(und), np.mean(disc), np.std(und), np.max(und), np.min(und)
    else:
        n, s1, s2, sd = sess.allreduce([len(und), np.sum(und), np.sum(und * und), np.sum(disc)])
        hi, neg_lo = sess.allreduce([np.max(und), -np.min(und)], 'max')
        n, mean, mean_disc, lo = int(round(n)), s1 / n, sd / n, -neg_lo
        std = np.sqrt(max(s2 / n - mean * mean, 0.0))
    if log == 'reward':
        logger.logkv(log_prefix + 'AverageReturn', mean)
        return
    logger.logkv(log_prefix + 'AverageDiscountedReturn', mean_disc)
    logger.logkv(log_prefix + 'AverageReturn', mean)
    logger.logkv(log_prefix + 'NumTrajs', n)
    logger.logkv(log_prefix + 'StdReturn', std)
    logger.logkv(log_prefix + 'MaxReturn', hi)
    logger.logkv(log_prefix + 'MinReturn', lo)
