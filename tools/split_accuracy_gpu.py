"""Developer tool: float32-equivalence of the BF16-split GEMMs as the kernels compute them, measured on the device against the
float64 oracle on small well-conditioned cases (the numbers tests/test_gpu_parity.py::test_split_gemm_accuracy_guard pins).
usage (GPU box): python tools/split_accuracy_gpu.py"""
import sys
import numpy as np
sys.path.insert(0, '.')
from oracle import policy as op, promp as pm
from promp_amd import _lib
from tests import helpers, parity_checks as pc

lib = _lib.Library(sys.argv[1]) if len(sys.argv) > 1 else _lib.Library()


def errs(hidden, O, A, seed=11, M=2, P=2, T=48):
    theta, all_slabs, all_paths = helpers.make_promp_case(seed, M, P, T, O, A, hidden, 1)
    spec = op.PolicySpec(O, A, hidden)
    ctx = pc.make_ctx(lib, M, O, A, hidden, 1, all_paths)
    helpers.upload_slabs(ctx, all_paths, all_slabs)
    rng = np.random.RandomState(seed + 1)
    th = (theta + 0.02 * rng.randn(M, theta.size)).astype(np.float32)
    ctx.set_task_thetas(th)
    out = {}
    for kind, name in ((0, 'ratio'), (2, 'loglik')):
        g, l, k = ctx.eval_loss_grad(1, kind, clip_eps=0.3, clip_log_std=False)
        out['grad_' + name] = max(pc.rel_max(g[i], pm.loss_and_grad(spec, th[i].astype(np.float64), all_slabs[1][i], name, False, clip_eps=0.3)['grad']) for i in range(M))
    v = rng.randn(M, theta.size).astype(np.float32)
    for kind, name in ((0, 'ratio'), (1, 'loglik')):
        hv = ctx.eval_hvp(0, v, inner_kind=kind, clip_log_std=True, kl_weight=0.37)
        e = 0
        for i in range(M):
            t64 = th[i].astype(np.float64)
            ref = -pm.hvp(spec, t64, all_slabs[0][i], v[i].astype(np.float64), name, True) + 0.37 * pm.loss_and_grad(spec, t64, all_slabs[0][i], name, True)['grad_kl']
            e = max(e, pc.rel_max(hv[i], ref))
        out['hvp_' + name] = e
    alpha = np.full(spec.n_params, 0.1, np.float32)
    eta = np.array([5e-4], np.float32)
    ctx.set_theta(theta)
    ctx.set_step_sizes(alpha)
    for cache in (0, 1):
        ctx.set_primal_cache(cache)
        g, st = ctx.meta_grad(0.3, eta)
        r = pm.meta_objective_and_grad(spec, theta.astype(np.float64), all_slabs, alpha.astype(np.float64), eta.astype(np.float64), 0.3)
        out['meta_cache%d' % cache] = pc.rel_max(g, r['grad'])
    ctx.close()
    return out


import os
GENERIC = (((64, 64, 64), 20, 6), ((256, 256), 20, 6), ((64, 64), 376, 17), ((100,), 11, 3))      # the layer-by-layer kernels
FUSED = (((64, 64), 20, 6), ((32, 32), 7, 3), ((32, 64), 20, 6), ((64, 32), 11, 2), ((128, 128), 111, 8), ((128, 128), 20, 6))
for hidden, O, A in (GENERIC if os.environ.get('PROMP_ACC_GENERIC_ONLY') else FUSED + GENERIC):
    print(hidden, O, A, {k: '%.2e' % v for k, v in errs(hidden, O, A).items()})
