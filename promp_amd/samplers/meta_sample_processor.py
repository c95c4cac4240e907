"""MetaSampleProcessor (reference: meta_policy_search/samplers/meta_sample_processor.py:8-49)."""
import numpy as np

from .base import SampleProcessor


class MetaSampleProcessor(SampleProcessor):

    def process_samples(self, paths_meta_batch, log=False, log_prefix=''):
        """
        Args:
            paths_meta_batch (dict): {meta_task -> list of path dicts}, size [meta_batch_size] x (batch_size) x [5] x (T)
        Returns:
            (list of dicts): processed sample data per meta task; 8 keys: observations, actions, rewards, returns,
            advantages, env_infos, agent_infos, adj_avg_rewards.  The trajectory slab and its advantages stay
            resident on the GPU for MAMLAlgo._adapt / ProMP.optimize_policy.
        """
        assert isinstance(paths_meta_batch, dict), 'paths must be a dict'
        assert self.baseline, 'baseline must be specified'
        samples_data_meta_batch, out = self._process_on_device(paths_meta_batch)
        # 7) normalized trajectory-batch rewards for E-MAML (meta_sample_processor.py:40-44), from the per-path sums
        n = sum(len(sd['rewards']) for sd in samples_data_meta_batch)
        mean = np.sum(out['path_undiscounted']) / n
        std = np.sqrt(max(np.sum(out['path_reward_sumsq']) / n - mean * mean, 0.0))
        for sd in samples_data_meta_batch:
            sd['adj_avg_rewards'] = (sd['rewards'] - mean) / (std + 1e-8)
        # 8) log statistics if desired
        self._log_path_stats(out, log=log, log_prefix=log_prefix)
        return samples_data_meta_batch
