"""DiagonalGaussian: NumPy-side helpers of the reference's distribution object
(meta_policy_search/policies/distributions/diagonal_gaussian.py:47-69,111-172).  The symbolic (*_sym) methods
of the reference are TF graph builders; their arithmetic lives in the HIP kernels."""
import numpy as np


class DiagonalGaussian(object):
    def __init__(self, dim):
        self._dim = dim

    @property
    def dim(self):
        return self._dim

    def kl(self, old_dist_info, new_dist_info):
        old_std, new_std = np.exp(old_dist_info['log_std']), np.exp(new_dist_info['log_std'])
        num = np.square(old_dist_info['mean'] - new_dist_info['mean']) + np.square(old_std) - np.square(new_std)
        den = 2 * np.square(new_std) + 1e-8
        return np.sum(num / den + new_dist_info['log_std'] - old_dist_info['log_std'], axis=-1)

    def log_likelihood(self, xs, dist_info):
        zs = (xs - dist_info['mean']) / np.exp(dist_info['log_std'])
        return -np.sum(dist_info['log_std'], axis=-1) - 0.5 * np.sum(np.square(zs), axis=-1) - 0.5 * self.dim * np.log(2 * np.pi)

    def entropy(self, dist_info):
        return np.sum(dist_info['log_std'] + np.log(np.sqrt(2 * np.pi * np.e)), axis=-1)

    def sample(self, dist_info):
        return np.random.normal(size=dist_info['mean'].shape) * np.exp(dist_info['log_std']) + dist_info['mean']

    @property
    def dist_info_specs(self):
        return [('mean', (self.dim,)), ('log_std', (self.dim,))]
