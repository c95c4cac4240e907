"""developer tool: the pair kernel against the one-wave-per-SIMD kernel on the same inputs (both on the GPU), per parameter block"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from promp_amd import _lib
import helpers, parity_checks as pc

def run(libpath, seed, M, P, T, O, A, hidden, reps):
    lib = _lib.Library(libpath)
    theta, all_slabs, all_paths = helpers.make_promp_case(seed, M, P, T, O, A, hidden, 1, ragged=True)
    ctx = pc.make_ctx(lib, M, O, A, hidden, 1, all_paths)
    helpers.upload_slabs(ctx, all_paths, all_slabs)
    rng = np.random.RandomState(seed + 1)
    th = (theta + 0.02 * rng.randn(M, theta.size)).astype(np.float32)
    ctx.set_task_thetas(th)
    out = []
    for rep in range(reps):
        for kind in (0, 1, 2):
            g, l, k = ctx.eval_loss_grad(1, kind, clip_eps=0.3, clip_log_std=False)
            out.append((g.copy(), l.copy(), k.copy()))
    ctx.close()
    return out

M, P, T, O, A, hidden = 5, 6, 150, 20, 6, (64, 64)
H1, H2 = hidden
blocks = [('W1', O * H1), ('b1', H1), ('W2', H1 * H2), ('b2', H2), ('W3', H2 * A), ('b3', A), ('ls', A)]
a = run(os.path.join(ROOT, 'tools/ablate/lib_nopair.so'), 31, M, P, T, O, A, hidden, 4)
b = run(os.path.join(ROOT, 'tools/ablate/lib_pair.so'), 31, M, P, T, O, A, hidden, 4)
for i, ((ga, la, ka), (gb, lb, kb)) in enumerate(zip(a, b)):
    line = 'eval %2d kind %d  loss %.2e kl %.2e |' % (i, i % 3, np.max(np.abs(la - lb)), np.max(np.abs(ka - kb)))
    o = 0
    for name, n in blocks:
        da = ga[:, o:o + n]; db = gb[:, o:o + n]
        e = np.max(np.abs(da - db), axis=1) / (np.max(np.abs(ga), axis=1) + 1e-30)
        line += ' %s %.1e(t%d)' % (name, e.max(), int(e.argmax()))
        o += n
    print(line)
# repeatability of the pair kernel itself
for i in range(3, len(b)):
    print('pair run-to-run eval', i, 'max |diff| vs first of its kind', float(np.max(np.abs(b[i][0] - b[i % 3][0]))))
# which half's entries are off (kind 2, first eval)
ga, gb = a[2][0], b[2][0]
for tsk in range(M):
    o = 0
    line = 'task %d:' % tsk
    for name, n in blocks:
        da = ga[tsk, o:o + n]; db = gb[tsk, o:o + n]
        sc = np.max(np.abs(ga[tsk])) + 1e-30
        if name == 'W1': e = np.abs(da - db).reshape(O, H1); parts = (e[:, :32].max(), e[:, 32:].max())
        elif name == 'W2': e = np.abs(da - db).reshape(H1, H2); parts = (e[:, :32].max(), e[:, 32:].max(), e[:32, :].max(), e[32:, :].max())
        elif name == 'W3': e = np.abs(da - db).reshape(H2, A); parts = (e[:32].max(), e[32:].max())
        elif name in ('b1', 'b2'): e = np.abs(da - db); parts = (e[:32].max(), e[32:].max())
        else: parts = (np.abs(da - db).max(),)
        line += ' %s[' % name + ' '.join('%.1e' % (p / sc) for p in parts) + ']'
        o += n
    print(line)
bad = np.argwhere(np.abs(ga - gb) > 1e-3 * np.max(np.abs(ga)))
print('entries off:', len(bad), 'of', ga.size, ' nan:', int(np.isnan(gb).sum()), ' inf:', int(np.isinf(gb).sum()))
