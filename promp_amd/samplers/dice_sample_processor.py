"""DiceSampleProcessor / DiceMetaSampleProcessor (reference: meta_policy_search/samplers/dice_sample_processor.py:6-230,
samplers/meta_sample_processor.py:50-51).

Contract: discounted rewards r_t gamma^t, a reward baseline fitted on them, adjusted rewards = discounted reward - baseline,
every path zero-padded to max_path_length with a mask, adjusted rewards normalised / shifted over the PADDED [paths,
max_path_length] array, path statistics logged from the discounted rewards.

Where the arithmetic runs: the baseline regression (features, normal equations, solve, predictions) is the device pipeline of
promp_process_samples with the regression target expressed as rewards whose undiscounted return IS the target
(r'_t = y_t - y_{t+1}, discount 1).  Padding, masks and the normalisation over the padded array are array bookkeeping on the
host.  The step's slab (observations, actions, agent_infos) stays resident on the device and receives the DiCE rewards of the
valid rows (promp_set_dice_rewards), so DICEMAML._adapt / optimize_policy need no second upload.
"""
from collections import OrderedDict

import numpy as np

from .. import _lib, session as session_mod
from .base import SampleProcessor, _concat_tensor_dict_list, _log_return_stats


class DiceSamplesData(session_mod.SamplesData):
    """padded samples of one task (mask, observations, actions, rewards, adjusted_rewards, env_infos, agent_infos)"""


class DiceSampleProcessor(SampleProcessor):
    """Args (dice_sample_processor.py:25-47): baseline, max_path_length, discount=0.99, gae_lambda=1, normalize_adv=True,
    positive_adv=False, return_baseline=None

    return_baseline (dice_sample_processor.py:113-124, 196-238): a second baseline, fitted on the returns, adds GAE
    'advantages' (padded, normalised / shifted over the padded array) beside the DiCE rewards -- what VPG_DICEMAML's outer
    objective reads.  Its regression and the GAE scan are a second run of the device pipeline on the same resident slab."""

    def __init__(self, baseline, max_path_length, discount=0.99, gae_lambda=1, normalize_adv=True, positive_adv=False,
                 return_baseline=None):
        assert 0 <= discount <= 1.0, 'discount factor must be in [0,1]'
        assert max_path_length > 0
        assert hasattr(baseline, 'fit') and hasattr(baseline, 'predict')
        super(DiceSampleProcessor, self).__init__(baseline, discount=discount, gae_lambda=gae_lambda, normalize_adv=normalize_adv,
                                                  positive_adv=positive_adv)
        self.max_path_length = max_path_length
        if return_baseline is not None:
            assert hasattr(return_baseline, 'fit') and hasattr(return_baseline, 'predict')
        self.return_baseline = return_baseline

    def _baseline_kind(self):
        """The regression runs on the device, so the baseline must be one the device knows (the `kind` of LinearFeatureBaseline /
        LinearTimeBaseline / ZeroBaseline).  Anything else would silently act as a zero baseline: refuse it instead."""
        kind = getattr(self.baseline, 'kind', None)
        if kind is None:
            raise TypeError('%s has no device-side `kind`: the DiCE sample processor fits its baseline on the device and takes '
                            'promp_amd.baselines.{LinearFeatureBaseline, LinearTimeBaseline, ZeroBaseline}' % type(self.baseline).__name__)
        return kind

    # -- padded arrays ---------------------------------------------------------------------------------------------------
    def _pad(self, array, path_length):
        array = np.asarray(array)
        assert path_length == array.shape[0]
        if array.ndim not in (1, 2):
            raise NotImplementedError
        return np.pad(array, ((0, self.max_path_length - path_length),) + ((0, 0),) * (array.ndim - 1), mode='constant')

    def _stack_padded(self, plist, key, sub=None):
        get = (lambda p: p[key][sub]) if sub is not None else (lambda p: p[key])
        return np.stack([self._pad(get(p), len(p['rewards'])) for p in plist], axis=0)

    def _process_meta_batch(self, paths_meta_batch):
        """-> (list of DiceSamplesData per task, all paths with 'discounted_rewards' / 'adjusted_rewards' added)"""
        T = self.max_path_length
        disc = np.cumprod(np.concatenate([np.ones(1), np.ones(T - 1) * self.discount]))
        shadow = OrderedDict()
        for i, plist in paths_meta_batch.items():
            sh = []
            for p in plist:
                n = len(p['rewards'])
                assert T >= n
                p['discounted_rewards'] = np.asarray(p['rewards']) * disc[:n]
                y = np.asarray(p['discounted_rewards'], dtype=np.float64)
                q = dict(p)
                q['rewards'] = y - np.append(y[1:], 0.0)            # undiscounted return of r' == the regression target
                sh.append(q)
            shadow[i] = sh
        M = len(paths_meta_batch)
        fl = _lib.flatten_paths(shadow)
        first = next(iter(paths_meta_batch.values()))[0]
        A = int(np.asarray(first['actions']).reshape(len(first['rewards']), -1).shape[1])
        sess = self._session_for(M, fl['obs'].shape[1], A)
        slot = sess.next_slot()
        upload = sess.upload_flat(slot, fl)
        ctx = sess.ctx
        kind = self._baseline_kind()
        ctx.process_samples(slot, discount=1.0, gae_lambda=1.0, normalize_adv=False, positive_adv=False, baseline_kind=kind,
                            reg_coeff=getattr(self.baseline, '_reg_coeff', 1e-5))
        if kind != _lib.BASELINE_ZERO:
            self.baseline._coeffs = ctx.download_processed(slot, kind)['coeffs'][-1].copy()
        base = ctx.predict_baseline(slot, kind)                    # Phi . w per valid row, float64
        pro, tpo = fl['path_row_offsets'], fl['task_path_offsets']
        gae = None
        if self.return_baseline is not None:
            # GAE advantages from a baseline fitted on the returns: the ordinary pipeline (returns, fit, predict, GAE) on the true
            # rewards of the same slab; normalisation happens below, over the padded array
            rkind = getattr(self.return_baseline, 'kind', None)
            if rkind is None:
                raise TypeError('%s has no device-side `kind`' % type(self.return_baseline).__name__)
            true_rewards = np.concatenate([np.asarray(p['rewards'], dtype=np.float64) for plist in paths_meta_batch.values() for p in plist])
            ctx.set_rewards(slot, true_rewards)
            ctx.process_samples(slot, discount=self.discount, gae_lambda=self.gae_lambda, normalize_adv=False, positive_adv=False,
                                baseline_kind=rkind, reg_coeff=getattr(self.return_baseline, '_reg_coeff', 1e-5))
            if rkind != _lib.BASELINE_ZERO:
                self.return_baseline._coeffs = ctx.download_processed(slot, rkind)['coeffs'][-1].copy()
            ret64, gae = ctx.download_raw(slot)
            for i, (_, plist) in enumerate(paths_meta_batch.items()):
                for j, p in enumerate(plist):
                    a, b = pro[tpo[i] + j], pro[tpo[i] + j + 1]
                    p['returns'], p['advantages'] = ret64[a:b], gae[a:b]
        result, rw_rows = [], []
        for i, (_, plist) in enumerate(paths_meta_batch.items()):
            for j, p in enumerate(plist):
                a, b = pro[tpo[i] + j], pro[tpo[i] + j + 1]
                p['adjusted_rewards'] = p['discounted_rewards'] - base[a:b]
            adj = self._stack_padded(plist, 'adjusted_rewards')
            if self.normalize_adv:
                adj = (adj - np.mean(adj)) / (np.std(adj) + 1e-8)   # over the padded array, zeros included (utils/utils.py:59-66)
            if self.positive_adv:
                adj = (adj - np.min(adj)) + 1e-8
            sd = DiceSamplesData(
                mask=np.stack([self._pad(np.ones(len(p['rewards'])), len(p['rewards'])) for p in plist], axis=0),
                observations=self._stack_padded(plist, 'observations'),
                actions=self._stack_padded(plist, 'actions'),
                rewards=self._stack_padded(plist, 'rewards'),
                env_infos={k: self._stack_padded(plist, 'env_infos', k) for k in plist[0].get('env_infos', {})},
                agent_infos={k: self._stack_padded(plist, 'agent_infos', k) for k in plist[0].get('agent_infos', {})},
                adjusted_rewards=adj)
            if gae is not None:
                adv = self._stack_padded(plist, 'advantages')
                if self.normalize_adv:
                    adv = (adv - np.mean(adv)) / (np.std(adv) + 1e-8)
                if self.positive_adv:
                    adv = (adv - np.min(adv)) + 1e-8
                sd['advantages'] = adv
            sd.device_ref = (sess.serial, upload, slot, i)
            result.append(sd)
            rows = sum(len(p['rewards']) for p in plist)
            scale = rows / float(len(plist) * T)                     # slab mean (1 / rows) -> mean over the padded array
            rw_rows.append(np.concatenate([adj[j, :len(p['rewards'])] for j, p in enumerate(plist)]) * scale)
        ctx.set_dice_rewards(slot, np.concatenate(rw_rows))
        return result, [p for plist in paths_meta_batch.values() for p in plist]

    # -- reference API ---------------------------------------------------------------------------------------------------
    def process_samples(self, paths, log=False, log_prefix=''):
        """single task: list of paths -> padded samples dict (dice_sample_processor.py:49-90)"""
        assert type(paths) == list, 'paths must be a list'
        assert paths[0].keys() >= {'observations', 'actions', 'rewards'}
        assert self.baseline, 'baseline must be specified - use self.build_sample_processor(baseline_obj)'
        result, all_paths = self._process_meta_batch(OrderedDict([(0, paths)]))
        self._log_dice_stats(all_paths, log=log, log_prefix='')   # the reference drops log_prefix here (:87)
        sd = dict(result[0])
        assert sd.keys() >= {'observations', 'actions', 'rewards', 'adjusted_rewards', 'mask'}
        return sd

    def _log_dice_stats(self, paths, log=False, log_prefix=''):
        """dice_sample_processor.py:133-147 ('discounted return' = the sum of a path's discounted rewards)"""
        _log_return_stats(self._stat_session(), [np.sum(p['rewards']) for p in paths],
                          [np.sum(p['discounted_rewards']) for p in paths], log, log_prefix)


class DiceMetaSampleProcessor(DiceSampleProcessor):
    """MetaSampleProcessor.process_samples on the DiCE processor (samplers/meta_sample_processor.py:8-51)"""

    def process_samples(self, paths_meta_batch, log=False, log_prefix=''):
        assert isinstance(paths_meta_batch, dict), 'paths must be a dict'
        assert self.baseline, 'baseline must be specified'
        samples_data_meta_batch, all_paths = self._process_meta_batch(paths_meta_batch)
        # rewards z-scored over the whole meta-batch (meta_sample_processor.py:40-44), here on the padded reward arrays
        overall = np.concatenate([sd['rewards'].reshape(-1) for sd in samples_data_meta_batch])
        sess = self._stat_session()
        if sess.world == 1:
            mean, std = np.mean(overall), np.std(overall)
        else:           # task-sharded: the padded arrays of the other ranks' tasks enter through their moments
            n, s1, s2 = sess.allreduce([overall.size, np.sum(overall, dtype=np.float64), np.sum(overall.astype(np.float64) ** 2)])
            mean = s1 / n
            std = np.sqrt(max(s2 / n - mean * mean, 0.0))
        for sd in samples_data_meta_batch:
            sd['adj_avg_rewards'] = (sd['rewards'] - mean) / (std + 1e-8)
        self._log_dice_stats(all_paths, log=log, log_prefix=log_prefix)
        return samples_data_meta_batch
