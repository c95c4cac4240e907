"""Developer tool: one ProMP iteration (process_samples x2, _adapt, 5 Adam epochs + stats) at policy shapes outside the fused
kernels -- the layer-by-layer kernels of promp_kernels_generic.h -- timed like bench.py's step, with the per-kernel profile.
usage: python tools/generic_timing.py [--steps N]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promp_amd import _lib, synthetic

CASES = [   # name, M, P, T, O, A, hidden, baseline
    ('three hidden layers (64,64,64), HalfCheetah data shapes', 40, 20, 100, 20, 6, (64, 64, 64), 'linear_feature'),
    ('(256,256), HalfCheetah data shapes', 40, 20, 100, 20, 6, (256, 256), 'linear_feature'),
    ('Humanoid dimensions (376 obs, 17 act), (64,64)', 40, 20, 200, 376, 17, (64, 64), 'linear_feature'),
    ('reference point: (64,64) on the fused kernels', 40, 20, 100, 20, 6, (64, 64), 'linear_feature'),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--case', type=int, default=-1, help='run only this entry of CASES')
    args = ap.parse_args()
    for name, M, P, T, O, A, hidden, baseline in (CASES if args.case < 0 else CASES[args.case:args.case + 1]):
        K, E, N = 1, 5, P * T
        theta0 = synthetic.init_theta(np.random.RandomState(1), O, hidden, A)
        ctx = _lib.Context(M, O, A, hidden, K, max_rows=M * N, max_paths=M * P)
        ctx.set_theta(theta0)
        ctx.set_step_sizes(np.full(ctx.n_params, 0.1, np.float32))
        opts = dict(discount=0.99, gae_lambda=1.0, normalize_adv=True, baseline_kind=dict(linear_feature=_lib.BASELINE_LINEAR_FEATURE, linear_time=_lib.BASELINE_LINEAR_TIME)[baseline])
        ids = list(range(M))
        f0 = _lib.flatten_paths(synthetic.make_paths_for_tasks(7, ids, theta0, P, T, O, A, hidden))
        ctx.upload_step(0, f0['task_path_offsets'], f0['path_row_offsets'], f0['obs'], f0['rew'], f0['act'], f0['old_mean'],
                        np.tile(theta0[-A:], (M, 1)))
        ctx.switch_to_pre_update()
        ctx.process_samples(0, **opts)
        ctx.inner_adapt(0)
        th1 = ctx.get_task_thetas()
        f1 = _lib.flatten_paths(synthetic.make_paths_for_tasks(8, ids, th1, P, T, O, A, hidden))
        ctx.upload_step(1, f1['task_path_offsets'], f1['path_row_offsets'], f1['obs'], f1['rew'], f1['act'], f1['old_mean'],
                        th1[:, -A:].copy())
        eta = np.array([5e-4], np.float32)

        def it():
            ctx.switch_to_pre_update()
            ctx.process_samples(0, **opts)
            ctx.inner_adapt(0)
            ctx.process_samples(1, **opts)
            ctx.optimize_begin(E, 1e-3, 0.3, eta)
            return ctx.optimize_end()
        for _ in range(2):
            it()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r = it()
        ctx.sync()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        print('%-62s Theta %6d  %8.3f ms/step  %7.1f M env-steps/s' % (name, ctx.n_params, ms, M * N * 2 / ms / 1e3), flush=True)
        ctx.prof_enable(True)
        it()
        ctx.sync()
        for kid, kname in ((0, 'pass (fwd + bwd)'), (1, 'R-operator pass'), (3, 'pass (fwd only)'), (2, 'gram')):
            v = ctx.prof_read(kid)
            if v['launches']:
                print('      %-18s %3d passes  %8.3f ms  (%.1f us each)' % (kname, v['launches'], v['total_ms'], 1e3 * v['total_ms'] / v['launches']))
        ctx.prof_enable(False)
        ctx.close()


if __name__ == '__main__':
    main()
