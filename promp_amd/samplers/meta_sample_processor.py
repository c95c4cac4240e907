"""MetaSampleProcessor (reference: meta_policy_search/samplers/meta_sample_processor.py:8-49)."""
import numpy as np

from .. import _lib
from .base import SampleProcessor


class MetaSampleProcessor(SampleProcessor):

    def process_samples(self, paths_meta_batch, log=False, log_prefix=''):
        """One sampling step of the whole meta-batch through returns, baseline fit, GAE and normalisation -- on the device.

        paths_meta_batch: dict {task index -> list of path dicts (observations, actions, rewards, env_infos, agent_infos)}.
        Returns a list with one dict per task holding the flattened observations, actions, rewards, returns, advantages,
        env_infos, agent_infos and adj_avg_rewards; as a side effect every path dict gains 'returns' and 'advantages'.
        The uploaded slab and its advantages stay resident for MAMLAlgo._adapt / optimize_policy (no second upload)."""
        assert isinstance(paths_meta_batch, dict), 'paths must be a dict'
        assert self.baseline, 'baseline must be specified'
        samples_data_meta_batch, out = self._process_on_device(paths_meta_batch)
        # rewards z-scored over the WHOLE meta-batch (E-MAML's exploration weight; meta_sample_processor.py:40-44 concatenates
        # every task's rewards), from the per-path sums the device returns.  A task-sharded run holds only this rank's tasks:
        # the three moments cross the ranks (one small all-reduce per sampling step).
        n, s1, s2 = self._stat_session().allreduce([sum(len(sd['rewards']) for sd in samples_data_meta_batch),
                                                    np.sum(out['path_undiscounted']), np.sum(out['path_reward_sumsq'])])
        mean = s1 / n
        std = np.sqrt(max(s2 / n - mean * mean, 0.0))
        # one subtraction and one division for the whole meta-batch; the per-task entries are slices of the result (the same values
        # as task by task, a tenth of the NumPy calls).  Where the per-row results are handed out lazily (a resident batch,
        # samplers/base.py) this one is too: only E-MAML's exploration term ever reads it.
        rew = [np.asarray(sd['rewards']) for sd in samples_data_meta_batch]
        compute = lambda: dict(adj=(np.concatenate(rew) - mean) / (std + 1e-8))
        if isinstance(samples_data_meta_batch[0]['advantages'], _lib.LazyRows):
            holder, dt = _lib.LazyCompute(compute), np.result_type(rew[0].dtype, np.float64)
            piece = lambda a, b: _lib.LazyRows(holder, 'adj', a, b, dtype=dt)
        else:
            adj = compute()['adj']
            piece = lambda a, b: adj[a:b]
        a = 0
        for sd, r in zip(samples_data_meta_batch, rew):
            sd['adj_avg_rewards'] = piece(a, a + len(r))
            a += len(r)
        self._log_path_stats(out, log=log, log_prefix=log_prefix)
        return samples_data_meta_batch
