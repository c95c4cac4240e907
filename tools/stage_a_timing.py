"""Developer tool: wall time of one process_samples(step 0) chain (returns, gram, fit, gae, normalize) on config-3 shapes,
back to back on the main stream (no second-stream overlap), for A/B runs of the sample-processing kernels.
usage: stage_a_timing.py [tasks [obs_dim]]   (obs_dim > 32: Ant shapes, k_gram_wide / k_fit_wide)"""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promp_amd import _lib, synthetic

M = int(sys.argv[1]) if len(sys.argv) > 1 else 40
O = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P, T, A, hidden = 20, 200, (6 if O <= 32 else 8), ((64, 64) if O <= 32 else (128, 128))
rng = np.random.RandomState(0)
theta = synthetic.init_theta(rng, O, hidden, A)
ctx = _lib.Context(M, O, A, hidden, 1, max_rows=M * P * T, max_paths=M * P)
p = synthetic.make_paths(rng, theta, M, P, T, O, A, hidden)
f = _lib.flatten_paths(p)
ctx.upload_step(0, f['task_path_offsets'], f['path_row_offsets'], f['obs'], f['rew'], f['act'], f['old_mean'], f['old_log_std'])
opts = dict(discount=0.99, gae_lambda=1.0, normalize_adv=True)
for _ in range(5):
    ctx.process_samples(0, **opts)
ctx.sync()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    ctx.process_samples(0, **opts)
ctx.sync()
print('M=%d O=%d process_samples(0): %.1f us' % (M, O, (time.perf_counter() - t0) / n * 1e6))
