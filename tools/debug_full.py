"""Developer tool (GPU box): gradient and Hessian-vector product of the fused kernels against the float64 oracle at full rows per task.
usage: python tools/debug_full.py [M] [P] [T]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from oracle import policy as op, promp as pm
from promp_amd import _lib
from tests import helpers, parity_checks as pc

M, P, T = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (8, 20, 200)
O, A, hidden = 20, 6, (64, 64)
lib = _lib.Library()
theta, all_slabs, all_paths = helpers.make_promp_case(5, M, P, T, O, A, hidden, 1)
spec = op.PolicySpec(O, A, hidden)
ctx = pc.make_ctx(lib, M, O, A, hidden, 1, all_paths)
helpers.upload_slabs(ctx, all_paths, all_slabs)
rng = np.random.RandomState(3)
th = (theta + 0.02 * rng.randn(M, theta.size)).astype(np.float32)
ctx.set_task_thetas(th)
for kind, name in ((0, 'ratio'), (2, 'loglik')):
    g, l, k = ctx.eval_loss_grad(1, kind, clip_eps=0.3, clip_log_std=False)
    for i in range(min(M, 4)):
        ref = pm.loss_and_grad(spec, th[i].astype(np.float64), all_slabs[1][i], name, False, clip_eps=0.3)
        print('grad', name, 'task', i, 'rel err %.2e' % pc.rel_max(g[i], ref['grad']), 'nan', int(np.isnan(g[i]).sum()), 'loss', l[i], ref['loss'])
    print('  tasks with NaN:', [i for i in range(M) if np.isnan(g[i]).any()])
for scale in (1.0, 1e-3):
    v = (scale * rng.randn(M, theta.size)).astype(np.float32)
    hv = ctx.eval_hvp(0, v, inner_kind=0, clip_log_std=True, kl_weight=0.37)
    for i in range(min(M, 3)):
        t64 = th[i].astype(np.float64)
        ref = -pm.hvp(spec, t64, all_slabs[0][i], v[i].astype(np.float64), 'ratio', True) + 0.37 * pm.loss_and_grad(spec, t64, all_slabs[0][i], 'ratio', True)['grad_kl']
        print('hvp scale', scale, 'task', i, 'rel err %.2e' % pc.rel_max(hv[i], ref), 'nan', int(np.isnan(hv[i]).sum()))
    print('  tasks with NaN:', [i for i in range(M) if np.isnan(hv[i]).any()])
g1, _, _ = ctx.eval_loss_grad(1, 1, clip_eps=0.3, clip_log_std=False)
for vv, nm in ((0.1 * g1, 'alpha * outer gradient'), (np.where(np.arange(theta.size) >= theta.size - A, 0, 0.1 * g1).astype(np.float32), 'same, log_std part zeroed')):
    print(nm, 'max |v| %.3e' % np.abs(vv).max(), 'max |v[log_std]| %.3e' % np.abs(vv[:, -A:]).max(), 'max |v[W1]| %.3e' % np.abs(vv[:, :O * 64]).max())
    for klw in (5e-4, 0.0):
        hv = ctx.eval_hvp(0, vv, inner_kind=0, clip_log_std=True, kl_weight=klw)
        i = 0
        t64 = th[i].astype(np.float64)
        ref = -pm.hvp(spec, t64, all_slabs[0][i], vv[i].astype(np.float64), 'ratio', True) + klw * pm.loss_and_grad(spec, t64, all_slabs[0][i], 'ratio', True)['grad_kl']
        print('  klw', klw, 'task 0 rel err %.2e' % pc.rel_max(hv[i], ref), 'tasks with NaN:', [i for i in range(M) if np.isnan(hv[i]).any()])
alpha, eta = np.full(spec.n_params, 0.1, np.float32), np.array([5e-4], np.float32)
ctx.set_theta(theta)
ctx.set_step_sizes(alpha)
r = pm.meta_objective_and_grad(spec, theta.astype(np.float64), all_slabs, alpha.astype(np.float64), eta.astype(np.float64), 0.3)
for cache in (0, 1):
    ctx.set_primal_cache(cache)
    g, st = ctx.meta_grad(0.3, eta)
    print('split events before meta', ctx.split_events()) if cache == 0 else None
    print('meta cache', cache, 'rel err %.2e' % pc.rel_max(g, r['grad']), 'nan', int(np.isnan(g).sum()), st, r['loss'])
print('split events', ctx.split_events())
ctx.close()
