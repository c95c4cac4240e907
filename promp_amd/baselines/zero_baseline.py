import numpy as np

from .. import _lib
from .base import Baseline


class ZeroBaseline(Baseline):
    """Dummy baseline (reference: meta_policy_search/baselines/zero_baseline.py)."""
    kind = _lib.BASELINE_ZERO

    def get_param_values(self, **kwargs):
        return None

    def set_param_values(self, value, **kwargs):
        pass

    def fit(self, paths, **kwargs):
        pass

    def predict(self, path):
        return np.zeros_like(path['rewards'])
