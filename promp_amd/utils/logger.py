"""Minimal key/value logger with the reference's call surface (meta_policy_search/utils/logger.py:184-197,
345-351,376-396): log, logkv, dumpkvs, getkvs, save_itr_params, configure.  Writes stdout and, when configured
with a directory, progress.csv with the same keys the reference emits."""
import csv
import os
import pickle
from collections import OrderedDict

_kvs = OrderedDict()
_dir = None
_snapshot_mode = 'last'
_snapshot_gap = 1
_csv_keys = None
_quiet = False


def configure(dir=None, format_strs=None, snapshot_mode='last', snapshot_gap=1, quiet=False):
    global _dir, _snapshot_mode, _snapshot_gap, _csv_keys, _quiet
    _dir, _snapshot_mode, _snapshot_gap, _csv_keys, _quiet = dir, snapshot_mode, snapshot_gap, None, quiet
    if dir:
        os.makedirs(dir, exist_ok=True)


def log(*args):
    if not _quiet:
        print(*args)


def logkv(key, val):
    _kvs[key] = val


def getkvs():
    return _kvs


def dumpkvs():
    global _csv_keys
    if not _quiet:
        w = max([len(k) for k in _kvs] + [1])
        for k, v in _kvs.items():
            print('%-*s | %s' % (w, k, ('%.6g' % v) if isinstance(v, float) else v))
        print('-' * (w + 16))
    if _dir:
        path = os.path.join(_dir, 'progress.csv')
        if _csv_keys is None:
            _csv_keys = list(_kvs.keys())
            with open(path, 'w', newline='') as f:
                csv.writer(f).writerow(_csv_keys)
        with open(path, 'a', newline='') as f:
            csv.writer(f).writerow([_kvs.get(k, '') for k in _csv_keys])
    out = OrderedDict(_kvs)
    _kvs.clear()
    return out


def save_itr_params(itr, params):
    """snapshot modes of utils/logger.py:376-396"""
    if not _dir or _snapshot_mode == 'none':
        return
    if _snapshot_mode == 'all':
        name = 'itr_%d.pkl' % itr
    elif _snapshot_mode in ('gap', 'last_gap'):
        if itr % _snapshot_gap != 0 and _snapshot_mode == 'gap':
            return
        name = 'itr_%d.pkl' % itr if itr % _snapshot_gap == 0 else 'params.pkl'
    else:
        name = 'params.pkl'
    with open(os.path.join(_dir, name), 'wb') as f:
        pickle.dump(params, f)
