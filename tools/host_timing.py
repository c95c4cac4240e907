"""Developer tool: host-side duration of every library call of one bench step (where does the host block?)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from promp_amd import _lib, synthetic

M, P, T, O, A, hidden = 40, 20, 200, 20, 6, (64, 64)
rng = np.random.RandomState(0)
theta = synthetic.init_theta(rng, O, hidden, A)
ctx = _lib.Context(M, O, A, hidden, 1, max_rows=M * P * T, max_paths=M * P)
ctx.set_theta(theta); ctx.set_step_sizes(np.full(ctx.n_params, 0.1, np.float32)); ctx.switch_to_pre_update()
for k in (0, 1):
    p = synthetic.make_paths(rng, theta, M, P, T, O, A, hidden)
    f = _lib.flatten_paths(p)
    ctx.upload_step(k, f['task_path_offsets'], f['path_row_offsets'], f['obs'], f['rew'], f['act'], f['old_mean'], f['old_log_std'])
opts = dict(discount=0.99, gae_lambda=1.0, normalize_adv=True)
eta = np.array([5e-4], np.float32)
calls = [('switch', lambda: ctx.switch_to_pre_update()), ('ps0', lambda: ctx.process_samples(0, **opts)), ('adapt', lambda: ctx.inner_adapt(0)),
         ('ps1', lambda: ctx.process_samples(1, **opts)), ('end', None), ('begin', lambda: ctx.optimize_begin(5, 1e-3, 0.3, eta))]
pending = False
rows = []
ctx.sync()
t_start = time.perf_counter()
for it in range(12):
    row = []
    for name, fn in calls:
        t0 = time.perf_counter()
        if name == 'end':
            if pending:
                ctx.optimize_end()
        else:
            fn()
        row.append((time.perf_counter() - t0) * 1e6)
    pending = True
    rows.append(row)
ctx.optimize_end()
ctx.sync()
print('total per step %.1f us' % ((time.perf_counter() - t_start) / 12 * 1e6))
print('      ' + ' '.join('%8s' % n for n, _ in calls))
for r in rows:
    print('      ' + ' '.join('%8.1f' % x for x in r))
