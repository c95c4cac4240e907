"""Host-side helpers the reference's run scripts, samplers and environments import from meta_policy_search.utils.utils
(utils/utils.py:43-190): the NumPy pieces under their own names, so that `from promp_amd.utils.utils import set_seed, ClassEncoder`
is the whole change in a run script (run_scripts/pro-mp_run_mujoco.py:9-10, 92-97).  The TensorFlow-name helpers of that module
(get_original_tf_name, remove_scope_from_name, create_feed_dict, ...) have no counterpart: there is no graph and no variable scope
here, parameters are addressed by the reference's names directly (policies/meta_gaussian_mlp_policy.py).

The sample processor does not call these on the hot path (returns, advantages and their normalisation run on the device:
promp_process_samples); they are here for user code -- custom sample processors, environments' log_diagnostics, plotting."""
import json
import random

import numpy as np


def extract(x, *keys):
    """values of `keys` from a dict, or per key the list of values over a list of dicts (utils.py:43-56)"""
    if isinstance(x, dict):
        return tuple(x[k] for k in keys)
    if isinstance(x, list):
        return tuple([item[k] for item in x] for k in keys)
    raise NotImplementedError('extract: a dict or a list of dicts, not %s' % type(x).__name__)


def normalize_advantages(advantages):
    """zero mean, unit standard deviation up to 1e-8 (utils.py:59-67)"""
    advantages = np.asarray(advantages)
    return (advantages - advantages.mean()) / (advantages.std() + 1e-8)


def shift_advantages_to_positive(advantages):
    """smallest entry becomes 1e-8 (utils.py:70-71)"""
    advantages = np.asarray(advantages)
    return (advantages - advantages.min()) + 1e-8


def discount_cumsum(x, discount):
    """y[t] = x[t] + discount * y[t + 1] along axis 0 (utils.py:74-81 runs scipy.signal.lfilter over the reversed sequence; the same
    recurrence written out -- float64 accumulation as lfilter's, one pass from the end)"""
    x = np.asarray(x)
    y = np.empty(x.shape, dtype=np.result_type(x.dtype, np.float64))
    carry = np.zeros(x.shape[1:], dtype=y.dtype)
    g = float(discount)
    for t in range(x.shape[0] - 1, -1, -1):
        carry = x[t] + g * carry
        y[t] = carry
    return y


def explained_variance_1d(ypred, y):
    """1 - Var[y - ypred] / Var[y]; a constant target counts as explained (1) unless the prediction varies (0) (utils.py:84-101)"""
    ypred, y = np.asarray(ypred), np.asarray(y)
    assert y.ndim == 1 and ypred.ndim == 1
    vary = np.var(y)
    if np.isclose(vary, 0):
        return 0 if np.var(ypred) > 0 else 1
    return 1 - np.var(y - ypred) / (vary + 1e-8)


def _merge(dicts, leaf):
    out = dict()
    for key, first in dicts[0].items():
        column = [d[key] for d in dicts]
        out[key] = _merge(column, leaf) if isinstance(first, dict) else leaf(column)
    return out


def concat_tensor_dict_list(tensor_dict_list):
    """list of (nested) dicts of arrays -> one dict, arrays concatenated along axis 0 (utils.py:104-121)"""
    return _merge(list(tensor_dict_list), np.concatenate)


def stack_tensor_dict_list(tensor_dict_list):
    """list of (nested) dicts of per-step values -> one dict of arrays with the list index as axis 0 (utils.py:124-141)"""
    return _merge(list(tensor_dict_list), np.asarray)


def set_seed(seed):
    """Seeds every generator the package draws from (utils.py:161-177 seeds random, NumPy and TensorFlow): Python's and NumPy's.
    The device-side exploration noise (Philox, promp_policy_step / promp_rollout_point_env) takes its per-rollout seed from
    NumPy's generator, the parameter initialisation too -- so a seeded run is reproducible end to end, which the reference's
    TensorFlow sampling was not."""
    seed %= 4294967294
    random.seed(seed)
    np.random.seed(seed)
    print('using seed %s' % (str(seed)))


class ClassEncoder(json.JSONEncoder):
    """json.dump(config, ..., cls=ClassEncoder) of the run scripts (utils.py:179-185): classes and callables in a config are written
    by name"""

    def default(self, o):
        if isinstance(o, type):
            return {'$class': o.__module__ + '.' + o.__name__}
        if callable(o):
            return {'function': o.__name__}
        return json.JSONEncoder.default(self, o)
