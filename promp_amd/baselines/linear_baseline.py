"""LinearFeatureBaseline / LinearTimeBaseline (reference: meta_policy_search/baselines/linear_baseline.py).

The objects carry the configuration (feature kind, reg_coeff) and the coefficient vector; the arithmetic
(features, normal equations, solve, prediction) is fused into the device sample-processing pipeline that
MetaSampleProcessor drives (k_gram / k_fit / k_gae in promp_amd/csrc/promp_kernels_sample.h).  After
process_samples, ``_coeffs`` holds the fit of the last task, as in the reference where one shared baseline
object is re-fit task by task (linear_baseline.py:70)."""
import atexit
import threading

import numpy as np

from .. import _lib
from .base import Baseline

# Standalone fit / predict run on a one-task context of their own.  Creating a context costs ~40 device allocations, three
# streams and their events -- far more than the one small solve it is created for -- so one is kept per observation width
# and grown geometrically (a baseline is typically asked to predict path after path).
_standalone = {}
_standalone_lock = threading.RLock()      # the shared contexts' step 0 is one upload / fit / download sequence at a time


def _close_standalone():
    with _standalone_lock:
        for ctx in _standalone.values():
            try:
                ctx.close()
            except Exception:
                pass
        _standalone.clear()


atexit.register(_close_standalone)


def _standalone_context(obs_dim, rows, paths):
    lib = _lib.get_library()
    key = (id(lib), obs_dim)
    ctx = _standalone.get(key)
    if ctx is None or ctx.dims.max_rows < rows or ctx.dims.max_paths < paths:
        cap = (max(rows, 2 * ctx.dims.max_rows), max(paths, 2 * ctx.dims.max_paths)) if ctx is not None else (max(rows, 1024), max(paths, 16))
        if ctx is not None:
            ctx.close()
        from .. import session as session_mod
        cur = session_mod.current()               # the device of the process's session (a rank's GPU), device 0 without one
        ctx = _lib.Context(1, obs_dim, 1, (32, 32), 1, max_rows=cap[0], max_paths=cap[1],
                           device_id=cur.device_id if cur is not None else 0)
        _standalone[key] = ctx
    return ctx


class LinearBaseline(Baseline):
    kind = None

    def __init__(self, reg_coeff=1e-5):
        self._coeffs = None
        self._reg_coeff = reg_coeff

    def get_param_values(self, **tags):
        return self._coeffs

    def set_params(self, value, **tags):
        self._coeffs = value

    def fit(self, paths, target_key='returns'):
        """Standalone fit: targets are taken from path[target_key].  The device pipeline regresses on the returns
        it computes itself, so an arbitrary target is expressed as rewards whose undiscounted return IS the
        target: r[t] = y[t] - y[t+1] (discount 1)."""
        assert all(target_key in p for p in paths)
        from collections import OrderedDict
        shadow = []
        for p in paths:
            y = np.asarray(p[target_key], dtype=np.float64)
            r = y - np.append(y[1:], 0.0)
            shadow.append(dict(observations=p['observations'], rewards=r))
        fl = _lib.flatten_paths(OrderedDict([(0, shadow)]))
        with _standalone_lock:
            self._fit_locked(fl, len(paths))

    def _fit_locked(self, fl, n_paths):
        ctx = _standalone_context(fl['obs'].shape[1], len(fl['rew']), n_paths)
        ctx.upload_step(0, fl['task_path_offsets'], fl['path_row_offsets'], fl['obs'], fl['rew'])
        ctx.process_samples(0, discount=1.0, gae_lambda=1.0, baseline_kind=self.kind, reg_coeff=self._reg_coeff)
        self._coeffs = ctx.download_processed(0)['coeffs'][0].copy()

    def predict(self, path):
        """Phi . w for one path, evaluated on the device (zeros when unfit, linear_baseline.py:31-32)."""
        n = len(path['observations'])
        if self._coeffs is None:
            return np.zeros(n)
        from collections import OrderedDict
        fl = _lib.flatten_paths(OrderedDict([(0, [dict(observations=path['observations'], rewards=np.zeros(n))])]))
        with _standalone_lock:
            ctx = _standalone_context(fl['obs'].shape[1], n, 1)
            ctx.upload_step(0, fl['task_path_offsets'], fl['path_row_offsets'], fl['obs'], fl['rew'])
            ctx.set_coeffs(0, self.kind, np.asarray(self._coeffs, dtype=np.float64).reshape(1, -1))
            return ctx.predict_baseline(0, self.kind)

    def log_diagnostics(self, paths, prefix):
        pass


class LinearFeatureBaseline(LinearBaseline):
    """b(o,t) = w . [clip(o,+-10), clip(o)^2, t/100, (t/100)^2, (t/100)^3, 1]  (linear_baseline.py:101-106)"""
    kind = _lib.BASELINE_LINEAR_FEATURE


class LinearTimeBaseline(LinearBaseline):
    """b(t) = w . [t/100, (t/100)^2, (t/100)^3, 1]  (linear_baseline.py:122-126)"""
    kind = _lib.BASELINE_LINEAR_TIME
