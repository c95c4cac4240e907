"""Shared helpers for tests: golden loaders and slab construction (test infrastructure)."""
import glob
import json
import os
from collections import OrderedDict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def sample_proc_cases():
    return sorted(os.path.basename(p)[len('sample_proc_'):-4] for p in glob.glob(os.path.join(GOLDEN, 'sample_proc_*.npz')))


def promp_cases():
    return sorted(os.path.basename(p)[len('promp_autograd_'):-4] for p in glob.glob(os.path.join(GOLDEN, 'promp_autograd_*.npz')))


def load_sample_proc(name):
    """-> (meta dict, paths_meta_batch OrderedDict, golden npz)."""
    g = np.load(os.path.join(GOLDEN, 'sample_proc_%s.npz' % name))
    meta = json.loads(str(g['meta']))
    lens = g['path_lengths']
    A = g['actions'].shape[1]
    paths, off = OrderedDict(), 0
    for i in range(lens.shape[0]):
        plist = []
        for L in lens[i]:
            sl = slice(off, off + int(L))
            plist.append(dict(observations=g['observations'][sl].copy(), actions=g['actions'][sl].copy(),
                              rewards=g['rewards'][sl].copy(), env_infos={},
                              agent_infos=dict(mean=np.zeros((int(L), A), np.float32),
                                               log_std=np.zeros((int(L), A), np.float32))))
            off += int(L)
        paths[i] = plist
    return meta, paths, g


def load_promp(name):
    """-> (case dict, theta float32, all_slabs list[K+1] of list[M] of slab dicts, golden npz)."""
    g = np.load(os.path.join(GOLDEN, 'promp_autograd_%s.npz' % name))
    c = json.loads(str(g['meta']))
    all_slabs = []
    for k in range(c['K'] + 1):
        slabs = []
        for i in range(c['M']):
            slabs.append(dict(observations=g['step%d_observations' % k][i], actions=g['step%d_actions' % k][i],
                              advantages=g['step%d_advantages' % k][i],
                              agent_infos=dict(mean=g['step%d_mean' % k][i], log_std=g['step%d_log_std' % k][i])))
        all_slabs.append(slabs)
    return c, g['theta'], all_slabs, g
