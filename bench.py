#!/usr/bin/env python
"""bench.py -- env-steps/s through GAE + inner adapt + outer ProMP update (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic meta-batch already resident in HBM:
    process_samples(step 0) -> _adapt -> process_samples(step 1) -> optimize_policy (E=5 Adam epochs + stats)
i.e. Trainer.train()'s timed stages Time-SampleProc + Time-InnerStep + Time-OuterStep (meta_trainer.py:96-142)
on BASELINE config 3: 40-task HalfCheetahRandVel shapes (obs 20, act 6, 2x64 tanh MLP, H=200, P=20, K=1).

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run, one rank per GPU)

Multi-GPU: tasks are sharded i -> GPU (i mod N); the only exchange is one RCCL all-reduce of [Theta+K+2] floats per Adam
epoch (+1 for the stats pass).  Default is STRONG scaling: the named 40-task batch split over the N ranks (5 tasks per GPU
at N = 8); the same line also carries "weak_batch": 40 tasks per GPU (meta_batch_size = 40 N), timed right after.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from promp_amd import _lib, comm, synthetic  # noqa: E402

FP32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32 MFMA == f32 vector peak
F16_PEAK_TFLOPS = 2516.6      # MI355X_MICROARCH.md: dense FP16 / BF16 MFMA peak (no sparsity)
SPLIT_PRODUCTS = 3            # matrix instructions per float32-equivalent product of the two-term FP16 split (promp_device.h)
HBM_PEAK_GBS = 8000.0
ENV_NAMES = {1: 'Point2D', 2: 'HalfCheetahRandVel', 3: 'HalfCheetahRandVel', 4: 'AntRandDirec', 5: 'HalfCheetahRandVel'}


def flops_per_row(O, H1, H2, A):
    """matmul FLOPs, 2/MAC.  FWD = 2(O*H1+H1*H2+H2*A); BWD = 2*FWD - 2*O*H1; HVP = 2*(FWD+BWD)."""
    fwd = 2 * (O * H1 + H1 * H2 + H2 * A)
    bwd = 2 * fwd - 2 * O * H1
    return dict(fwd=fwd, bwd=bwd, fwd_bwd=fwd + bwd, hvp=2 * (fwd + bwd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=3, choices=[1, 2, 3, 4, 5],
                    help='BASELINE.json config; 3 is the one the metric is quoted on; 5 = config 3 shapes with the TRPO-MAML outer step '
                         '(conjugate gradients + line search instead of E Adam epochs); the others are shape studies')
    ap.add_argument('--scaling', default='strong', choices=['strong', 'weak'],
                    help='strong (default): the named 40-task config sharded over the N GPUs; weak: the named config per GPU')
    ap.add_argument('--epochs', type=int, default=5)
    ap.add_argument('--shard-of', type=int, default=0, metavar='N',
                    help='developer option: with --gpus 1, run only the shard rank 0 of an N-GPU job would hold '
                         '(M/N tasks, no collective) to study the per-rank step time of strong scaling')
    ap.add_argument('--repeats', type=int, default=0,
                    help='timed loops of exactly --steps steps each; `value` is the MEDIAN loop (default: 5 when one loop is '
                         'shorter than half a second, else 1)')
    ap.add_argument('--no-plugin-path', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--force-primal-cache', action='store_true',
                    help='developer switch: promp_set_primal_cache(1) (the default since round 6; through round 5 the default was on from two rounds of tiles per compute unit)')
    ap.add_argument('--no-primal-cache', action='store_true',
                    help='developer switch: the second-order pass recomputes the activations instead of reading the gradient '
                         'pass\'s copies back (promp_set_primal_cache)')
    ap.add_argument('--schedule', default='', metavar='OVERLAP,FUSE_MIN_TASKS',
                    help='developer switch: promp_set_schedule(stage_overlap, fuse_min_tasks), e.g. "0,-1" = all sample processing '
                         'on the main stream (results do not depend on it)')
    ap.add_argument('--staged-only', action='store_true',
                    help='developer switch: feed the timed loop by staged uploads (tools/gpu_round.sh traces the copy / compute overlap with it)')
    ap.add_argument('--test-shape', default='', metavar='M,P,T,O,A,H',
                    help='test switch: a tiny meta-batch instead of the named config (tests/test_bench_ranks.py runs the N-rank code path '
                         'of this file on the kernel emulator); the line it prints is marked and is not a benchmark result')
    args = ap.parse_args()

    rank, world, local_rank = comm.env_world()
    if world != args.gpus and world == 1 and 'RANK' not in os.environ:
        # `python bench.py --gpus N` with no launcher: start the N ranks ourselves (promp_amd.launch: one process per GPU, RANK /
        # LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run sets them) and let rank 0 print the line
        import socket
        from promp_amd import launch
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        sys.exit(launch.main(['--nproc', str(args.gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
                              os.path.abspath(sys.argv[0])] + sys.argv[1:]))
    if world != args.gpus:
        print('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    trpo = args.config == 5
    cfg = synthetic.CONFIGS[3 if trpo else args.config]
    if args.test_shape:
        tm, tp, tt, to, ta, thh = [int(x) for x in args.test_shape.split(',')]
        cfg = dict(M=tm, P=tp, T=tt, O=to, A=ta, hidden=(thh, thh))
    P, T, O, A, hidden = cfg['P'], cfg['T'], cfg['O'], cfg['A'], cfg['hidden']
    K, E = 1, args.epochs
    N = P * T
    opts = dict(discount=0.99, gae_lambda=1.0, normalize_adv=True)
    eta = np.array([5e-4], np.float32)
    seed = 1000 * args.config
    theta0 = synthetic.init_theta(np.random.RandomState(seed), O, hidden, A)

    def setup(M_global):
        """One resident meta-batch of M_global tasks, task i on rank i mod world; returns (ctx, iteration, M_local)."""
        task_ids = [i for i in range(M_global) if i % world == rank]
        if args.shard_of > 1 and world == 1:
            task_ids = [i for i in range(M_global) if i % args.shard_of == 0]
        M = len(task_ids)
        # (--shard-of: one rank's share timed alone, no collective -- the context is told its tasks are the whole batch, or the
        #  library would refuse to apply a shard's sums as the meta-batch's; only the 1 / M scaling differs, not the work)
        ctx = _lib.Context(M, O, A, hidden, K, max_rows=M * N, max_paths=M * P,
                           n_tasks_global=M if (args.shard_of > 1 and world == 1) else M_global, device_id=local_rank)
        if args.no_primal_cache:
            ctx.set_primal_cache(False)
        if args.force_primal_cache:
            ctx.set_primal_cache(True)
        if args.schedule:
            ctx.set_schedule(*[int(x) for x in args.schedule.split(',')])
        if world > 1:
            uid = comm.exchange_unique_id(rank, world, lambda: _lib.comm_unique_id())
            ctx.comm_init(rank, world, uid)
            if os.environ.get('PROMP_FIXED_ORDER', '0') not in ('', '0'):
                ctx.comm_fixed_order(True)
        # ---- synthetic, seeded, resident in HBM before timing ----
        ctx.set_theta(theta0)
        ctx.set_step_sizes(np.full(ctx.n_params, 0.1, np.float32))
        p0 = synthetic.make_paths_for_tasks(seed, task_ids, theta0, P, T, O, A, hidden)
        f0 = _lib.flatten_paths(p0)
        ctx.upload_step(0, f0['task_path_offsets'], f0['path_row_offsets'], f0['obs'], f0['rew'], f0['act'], f0['old_mean'],
                        np.tile(theta0[-A:], (M, 1)))
        ctx.switch_to_pre_update()
        ctx.process_samples(0, **opts)
        ctx.inner_adapt(0)
        th1 = ctx.get_task_thetas()           # post-update policies "sample" step 1 (ratio == 1 at the first epoch)
        p1 = synthetic.make_paths_for_tasks(seed + 1, task_ids, th1, P, T, O, A, hidden)
        f1 = _lib.flatten_paths(p1)
        ctx.upload_step(1, f1['task_path_offsets'], f1['path_row_offsets'], f1['obs'], f1['rew'], f1['act'], f1['old_mean'],
                        th1[:, -A:].copy())

        def upload():      # what a host-side sampler pays on top of the timed path: both slabs over PCIe (pageable memory)
            ctx.upload_step(0, f0['task_path_offsets'], f0['path_row_offsets'], f0['obs'], f0['rew'], f0['act'], f0['old_mean'],
                            np.tile(theta0[-A:], (M, 1)))
            ctx.upload_step(1, f1['task_path_offsets'], f1['path_row_offsets'], f1['obs'], f1['rew'], f1['act'], f1['old_mean'],
                            th1[:, -A:].copy())
            ctx.sync()

        if trpo:
            # TRPOMAML.optimize_policy (meta_algos/trpo_maml.py:161-191) on the same device passes: KL before, loss before,
            # ConjugateGradientOptimizer.optimize (10 CG iterations on finite-difference HVPs of the constraint, backtracking
            # line search), loss after, KL after; inner objective = log-likelihood (run_scripts/maml_run_mujoco.py:119)
            from types import SimpleNamespace
            from promp_amd.meta_algos.trpo_maml import _DeviceEvaluator
            from promp_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer
            from promp_amd.utils import logger as plog
            plog.configure(quiet=True)
            shim = SimpleNamespace(session=SimpleNamespace(ctx=ctx, M_global=M_global, world=world, external=lambda: False,
                                                           meta_eval=lambda *a, **k: ctx.meta_eval(*a, **k)), num_inner_grad_steps=K,
                                   inner_kind=_lib.INNER_LOGLIK, exploration=False, meta_batch_size=M)
            from promp_amd.optimizers.conjugate_gradient_optimizer import ExactDeviceHvp
            cgs = dict(finite_difference=ConjugateGradientOptimizer(), exact=ConjugateGradientOptimizer(hvp_approach=ExactDeviceHvp()),
                       host_loop=ConjugateGradientOptimizer(device_solve=False))
            for o in cgs.values():
                o.build_graph(_DeviceEvaluator(shim), 0.01)
            mode = dict(hvp='finite_difference')     # the reference's construction is the timed default (parity mode)

        def iteration(deferred=False):
            ctx.switch_to_pre_update()                       # meta_trainer.py:85
            ctx.process_samples(0, **opts)                   # :105  (step 0)
            ctx.inner_adapt(0, _lib.INNER_LOGLIK if trpo else _lib.INNER_RATIO)     # :116
            ctx.process_samples(1, **opts)                   # :105  (step 1)
            if trpo:
                cg = cgs[mode['hvp']]
                if cg._ev.objectives_ride_on_gradient():      # as TRPOMAML.optimize_policy: the loss gradient's pass yields both values
                    cg.gradient()
                kl0, l0 = cg.constraint_val(), cg.loss()
                cg.optimize()
                return dict(loss_before=l0, loss_after=cg.loss(), kl_before=kl0, kl_after=cg.constraint_val(),
                            n_backtracks=cg.last['n_backtracks'], rejected=cg.last['rejected'])
            # :128  E Adam epochs + compute_stats.  The statistics of iteration n are collected (promp_optimize_end) in front of
            # the optimisation of iteration n+1, where the KL-coefficient rule needs them (pro_mp.py:201-214): the host is
            # not held at the end of the step and enqueues the next batch's sample processing while this one is optimised.
            # run_timed() collects the last one inside the timed region.
            return optimize(deferred)
        pending = {'on': False}

        def optimize(deferred):
            prev = ctx.optimize_end() if pending['on'] else None
            ctx.optimize_begin(E, 1e-3, 0.3, eta)
            pending['on'] = deferred
            return prev if deferred else ctx.optimize_end()

        def collect():
            if pending['on']:
                pending['on'] = False
                return ctx.optimize_end()
            return None
        iteration.collect = collect
        def staged_iteration_factory():
            """the same iteration fed by staged uploads: the batch of iteration n+1 (here: the same synthetic batch again)
            leaves pinned host memory on the copy stream while iteration n is being optimised"""
            lib = ctx.lib
            pin = []
            for f, ls in ((f0, np.tile(theta0[-A:], (M, 1))), (f1, th1[:, -A:].copy())):
                d = {}
                for key, src in (('obs', f['obs']), ('rew', f['rew']), ('act', f['act']), ('old_mean', f['old_mean']), ('ls', ls)):
                    src = np.ascontiguousarray(src, dtype=np.float32)
                    d[key] = _lib.pinned_empty(lib, src.shape)
                    d[key][...] = src
                d['tpo'], d['pro'] = f['task_path_offsets'], f['path_row_offsets']
                pin.append(d)
            stage = lambda k: ctx.stage_step(k, pin[k]['tpo'], pin[k]['pro'], pin[k]['obs'], pin[k]['rew'], pin[k]['act'],
                                             pin[k]['old_mean'], pin[k]['ls'])
            stage(0), stage(1)

            def it(deferred=False):
                ctx.commit_step(0), ctx.commit_step(1)
                ctx.switch_to_pre_update()
                ctx.process_samples(0, **opts)
                ctx.inner_adapt(0, _lib.INNER_LOGLIK if trpo else _lib.INNER_RATIO)
                ctx.process_samples(1, **opts)
                stage(0), stage(1)                               # next batch: copy stream, under the epochs below
                return optimize(deferred)
            it.collect = collect
            return it
        iteration.upload = upload
        iteration.staged = staged_iteration_factory
        iteration.mode = mode if trpo else None
        iteration.reset = lambda: ctx.set_theta(theta0)
        return ctx, iteration, M

    def run_timed(ctx, iteration, warmup, n):
        """warmup untimed steps, then exactly n steps between barrier + device sync on both sides; max over ranks."""
        def barrier():
            ctx.sync()
            ctx.allreduce_f64([0.0])
            ctx.sync()
        res = None
        for _ in range(warmup):
            iteration()
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            r = iteration(deferred=True)
            res = r if r is not None else res
        r = iteration.collect()          # the last step's statistics: waited for inside the timed region
        res = r if r is not None else res
        ctx.sync()
        local = time.perf_counter() - t0     # this rank's own loop (its kernels + its waits for the exchanges)
        barrier()
        dt = time.perf_counter() - t0
        run_timed.local_s = local
        return float(ctx.allreduce_f64([dt], op='max')[0]), res

    M_global = cfg['M'] * (world if args.scaling == 'weak' else 1)
    ctx, iteration, M = setup(M_global)
    info = ctx.device_info()
    alpha = np.full(ctx.n_params, 0.1, np.float32)
    if args.staged_only:
        staged_it = iteration.staged()
        elapsed, res = run_timed(ctx, staged_it, args.warmup, args.steps)
        ctx.stage_wait()
    else:
        elapsed, res = run_timed(ctx, iteration, args.warmup, args.steps)
    # A loop of K steps is tens of milliseconds at config 3: one loop is one sample of the box's clocks.  Every repeat is the
    # contract's measurement again -- exactly K steps between barrier + device sync, max over ranks -- and `value` is the median
    # loop (all of them are listed).  The number of repeats is agreed on by the ranks (rank-local clocks differ).
    loops = [elapsed]
    n_rep = args.repeats if args.repeats > 0 else (5 if ctx.allreduce_f64([elapsed], op='max')[0] < 0.5 else 1)
    if not args.staged_only:
        for _ in range(n_rep - 1):
            el, r = run_timed(ctx, iteration, 0, args.steps)
            loops.append(el)
            res = r if r is not None else res
        elapsed = float(np.median(loops))
    if not np.isfinite(res['loss_after']):
        raise SystemExit('bench: non-finite loss')
    split_events = dict(ctx.split_events(), steps=args.warmup + args.steps * len(loops))      # over the warm-up and every timed loop
    trpo_last = None
    if trpo and res is not None:       # what the last timed step's trust-region step did (the same resident batch every step: see the note)
        trpo_last = {'n_backtracks': int(res.get('n_backtracks', -1)), 'rejected': bool(res.get('rejected', False)),
                     'kl_before': float(res['kl_before']), 'kl_after': float(res['kl_after']),
                     'note': 'the resident batch is the same every step: after the first accepted steps the policy sits at the edge of the '
                             'trust region around the distribution that sampled it, and a timed step then runs the line search to its '
                             'last candidate (max_backtracks = 15 evaluations) -- the search at its most expensive, in the host-loop '
                             'figure alike'}
    env_steps = M_global * N * (K + 1) * args.steps
    value = env_steps / elapsed

    out = {
        'metric': 'env-steps/sec through GAE+inner+outer update', 'value': value, 'unit': 'env-steps/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'timed_loops_ms_per_step': [1e3 * e / args.steps for e in loops],
        'fp16_split_events': split_events,
        **({'trpo_last_step': trpo_last} if trpo_last else {}),
        'config': {'workload': 'BASELINE config %d: %d-task %s shapes (obs=%d, act=%d, 2x%d tanh MLP, H=%d, '
                               'P=%d paths/task, K=1 inner step, %s); process_samples x2 + _adapt + '
                               'optimize_policy per step' % (args.config, M_global, ENV_NAMES[args.config], O, A, hidden[0], T, P,
                                                             'TRPO-MAML outer step: 10 CG iterations on finite-difference HVPs + line search' if trpo
                                                             else 'E=%d ProMP epochs + stats' % E),
                   'meta_batch_size': M_global, 'tasks_per_gpu': M, 'rows_per_task_per_step': N,
                   'env_steps_per_step': M_global * N * (K + 1), 'parallelism': 'task-sharded dp%d, RCCL all-reduce of the meta-gradient' % world,
                   'device': info['name'],
                   'numerics': ('float32-equivalent: 128-wide layers (k_wb_fwd_bwd, k_wb_hvp, promp_kernels_wide_bf16.h) compute every large GEMM as '
                                '6 BF16 products of a 3-way error-compensated split with float32 accumulation (measured <= 1.7e-6 of the float64 '
                                'oracle on gradient and Hessian-vector product, the exact-FP32 kernels\' own level; guarded by '
                                'test_split_gemm_accuracy_guard); sample processing float64') if (hidden[0] == 128 and O <= 127) else
                               'float32-equivalent: the first-order passes (k_pass) compute every GEMM, and the second-order pass '
                               '(k_chain_hvp) its layer 2, backward product and hidden_1 kernel gradient, as 3 FP16 products of a '
                               'two-term error-compensated split (hi.hi + hi.lo + lo.hi, float32 accumulation, operands kept inside '
                               'FP16\'s range by exact powers of two that follow the data: fp16_split_events counts the tiles / segments '
                               'walked again for it); measured <= 1.2e-6 of the float64 oracle, the FP32 fma chain\'s own level, '
                               'guarded at 2.5e-6 by test_split_gemm_accuracy_guard; the other GEMMs of the second-order pass on the '
                               'exact-FP32 MFMA; sample processing float64',
                   'schedule': 'the first Adam epoch takes its first inner pass (theta on step 0) from the _adapt call that just '
                               'evaluated it instead of repeating it (promp_set_reuse_adapt, default on; bit-identical, '
                               'test_first_epoch_reuses_the_inner_adapt_pass): 11 instead of 12 first-order passes per step'},
    }

    if args.test_shape:
        out['test_shape'] = 'NOT A BENCHMARK RESULT: --test-shape %s (a code-path test of this file)' % args.test_shape
    # ---- N ranks: who took part (what the communicator itself reports), every rank's own loop time, the exchange's latency ----
    if world > 1:
        ci = ctx.comm_info()
        bus = ci['pci_bus_id'].encode()[:32]
        row = np.zeros((world, 35))
        row[rank, 0] = 1e3 * run_timed.local_s / args.steps
        row[rank, 1] = ci['nranks']
        row[rank, 2] = ci['rank']
        row[rank, 3:3 + len(bus)] = list(bus)
        flat = row.reshape(-1)
        row = np.concatenate([ctx.allreduce_f64(flat[i:i + 64]) for i in range(0, flat.size, 64)]).reshape(world, 35)
        ctx.prof_enable(True)                         # HIP events around every exchange on the stream it is enqueued on
        n_ex = 1 if args.test_shape else 3
        for _ in range(n_ex):
            iteration()
        ex = ctx.prof_read(_lib.KERNEL_EXCHANGE)
        ctx.prof_enable(False)
        ex_us = ctx.allreduce_f64([1e3 * ex['total_ms'] / max(ex['launches'], 1)], op='max')[0]
        ideal = shard_ideal(args.config, world) if args.scaling == 'strong' and not args.test_shape else None
        # the replicas after the timed loops: every rank holds the same meta-parameters and Adam moments, or the exchange went wrong
        # (a checksum of each, min and max over the ranks; fixed-order exchange: bit-identical, all-reduce: equal to rounding)
        th = ctx.get_theta().astype(np.float64)
        m_, v_, t_ = ctx.get_adam_state()
        sums = [float(th.sum()), float(np.abs(th).sum()), float(m_.astype(np.float64).sum()), float(v_.astype(np.float64).sum()), float(t_)]
        lo, hi = -ctx.allreduce_f64([-x for x in sums], op='max'), ctx.allreduce_f64(sums, op='max')
        spread = max(abs(h - l) / max(abs(h), abs(l), 1e-300) for l, h in zip(lo, hi))
        out['rccl'] = {
            'replicas_equal': bool(spread <= (0.0 if ci['fixed_order'] else 1e-6)), 'replica_checksum_spread': float(spread),
            'replica_checksums': {'theta_sum': [float(lo[0]), float(hi[0])], 'theta_abs_sum': [float(lo[1]), float(hi[1])],
                                  'adam_m_sum': [float(lo[2]), float(hi[2])], 'adam_v_sum': [float(lo[3]), float(hi[3])],
                                  'adam_t': [float(lo[4]), float(hi[4])]},
            'nranks': int(row[0, 1]), 'nranks_reported_by_every_rank': [int(x) for x in row[:, 1]],
            'ranks': [int(x) for x in row[:, 2]],
            'devices': [bytes(int(c) for c in r[3:] if c > 0).decode() for r in row],
            'distinct_devices': len(set(bytes(int(c) for c in r[3:] if c > 0) for r in row)),
            'exchange': 'ncclAllGather + rank-ordered sum' if ci['fixed_order'] else 'ncclAllReduce(sum)',
            'exchange_floats': int(ctx.n_params + K + 2), 'exchanges_per_step': ex['launches'] / float(n_ex),
            'allreduce_us': ex_us, 'allreduce_us_note': 'HIP events around the exchange on the compute stream, mean per call, max over ranks '
                                                       '(includes waiting for the slowest rank to arrive)',
            'protocol': os.environ.get('NCCL_PROTO', 'RCCL default (LL for a %d-byte message)' % (4 * (ctx.n_params + K + 2))),
            'per_rank_ms_per_step': [float(x) for x in row[:, 0]]}
        if ideal:
            # what the committed one-GPU shard timing predicts for this run: the shard's kernels + one exchange per epoch and one for
            # the statistics at the latency measured right here
            n_exch = ex['launches'] / float(n_ex)
            predicted = ideal['shard_ms_per_step_kernels_only'] + 1e-3 * ex_us * n_exch
            out['strong_scaling'] = dict(ideal, speedup_vs_shard_ideal=ideal['shard_ms_per_step_kernels_only'] / (1e3 * elapsed / args.steps),
                                         measured_speedup_vs_profiled_single_gpu=ideal['single_gpu_ms_per_step'] / (1e3 * elapsed / args.steps),
                                         predicted_ms_per_step=predicted, predicted_speedup=ideal['single_gpu_ms_per_step'] / predicted,
                                         prediction='shard kernels %.3f ms + %.1f exchanges x %.1f us measured in this run'
                                                    % (ideal['shard_ms_per_step_kernels_only'], n_exch, ex_us))

    # ---- config 5 with the exact constraint Hessian-vector product (promp_constraint_hvp) instead of the reference's finite
    # difference: reported beside the timed default, never as `value` ----
    if trpo and rank == 0 and world == 1 and iteration.mode is not None:
        iteration.reset()
        iteration.mode['hvp'] = 'exact'
        n_ex = max(2, min(args.steps, 5))
        el_ex, res_ex = run_timed(ctx, iteration, 1, n_ex)
        iteration.mode['hvp'] = 'finite_difference'
        iteration.reset()
        iteration()
        out['exact_hvp'] = {'ms_per_step': 1e3 * el_ex / n_ex, 'value': M_global * N * (K + 1) * n_ex / el_ex, 'steps': n_ex,
                            'loss_after': float(res_ex['loss_after']), 'kl_after': float(res_ex['kl_after']),
                            'note': 'same step with hvp_approach=exact: 3 R-operator passes per product instead of two '
                                    'displaced constraint gradients (6 passes)'}
        # ... and with the optimizer looping over its products on the host (the reference's control flow: every displaced
        # gradient crosses PCIe and back) instead of promp_cg_solve's back-to-back launches
        iteration.mode['hvp'] = 'host_loop'
        el_h, res_h = run_timed(ctx, iteration, 1, n_ex)
        iteration.mode['hvp'] = 'finite_difference'
        iteration.reset()
        iteration()
        out['host_cg_loop'] = {'ms_per_step': 1e3 * el_h / n_ex, 'value': M_global * N * (K + 1) * n_ex / el_h, 'steps': n_ex,
                               'note': 'same step with ConjugateGradientOptimizer(device_solve=False): the conjugate-gradient loop on the '
                                       'host, 22 constraint gradients fetched one at a time'}

    # ---- host -> device cost of the two slabs (never part of `value`: the timed region starts with the batch in HBM) ----
    if rank == 0 and world == 1 and not args.staged_only:
        iteration.upload()
        t0 = time.perf_counter()
        for _ in range(3):
            iteration.upload()
        h2d_ms = 1e3 * (time.perf_counter() - t0) / 3
        step_ms = 1e3 * elapsed / args.steps
        out['h2d'] = {'upload_ms_per_step': h2d_ms, 'bytes_per_step': int(4 * M * N * 2 * (O + 2 * A + 1)),
                      'value_including_upload': M_global * N * (K + 1) / ((step_ms + h2d_ms) * 1e-3),
                      'note': 'promp_upload_step of both slabs from pageable host memory, incl. the host-side work tables, '
                              'serial with the step'}
        iteration()       # (re-create the processed state the uploads reset)
        if not trpo:
            # staged uploads from pinned memory, double-buffered slabs: the transfer of the next batch under this one's epochs
            it2 = iteration.staged()
            iteration.reset()
            n_st = max(3, min(args.steps, 20))
            el_st, _ = run_timed(ctx, it2, 2, n_st)
            ctx.stage_wait()
            out['h2d']['staged'] = {'ms_per_step': 1e3 * el_st / n_st, 'steps': n_st,
                                    'value_including_upload': M_global * N * (K + 1) * n_st / el_st,
                                    'note': 'promp_stage_step (pinned host arrays, copy stream) + promp_commit_step: every step '
                                            'uploads both slabs of the next batch while the current one is optimised'}
            iteration.reset()
            iteration.upload()
            iteration()

    # ---- roofline of the dominant kernel: HIP events around every launch, on the stream it runs on ----
    if not args.no_roofline:
        ctx.prof_enable(True)
        n_prof = max(3, min(args.steps, 10))
        for _ in range(n_prof):
            iteration()
        fl = flops_per_row(O, hidden[0], hidden[1], A)
        kern = {}
        chain = hidden[0] <= 64 and O <= 32
        wb = (not chain) and hidden[0] == 128 and O <= 127 and os.environ.get('PROMP_WIDE_FP32', '0') in ('', '0')   # promp_kernels_wide_bf16.h
        hvp_name = 'k_chain_hvp' if chain else 'k_wb_hvp' if wb else 'k_wide_hvp'
        pass_name = 'k_pass' if chain else 'k_wb_fwd_bwd' if wb else 'k_wide_fwd_bwd'
        for name, kid, f in ((pass_name, _lib.KERNEL_FWD_BWD, fl['fwd_bwd']), (hvp_name, _lib.KERNEL_HVP, fl['hvp']),
                             (pass_name + '<fwd-only>', _lib.KERNEL_FWD, fl['fwd'])):
            pr = ctx.prof_read(kid)
            avg_ms = pr['total_ms'] / max(pr['launches'], 1)
            rows = pr['rows'] / max(pr['launches'], 1)
            kern[name] = dict(avg_ms=avg_ms, launches_per_step=pr['launches'] / n_prof, rows_per_launch=rows,
                              flop_per_row=f, tflops=f * rows / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0,
                              ms_per_step=pr['total_ms'] / n_prof)
        pg = ctx.prof_read(_lib.KERNEL_GRAM)
        kern['k_gram'] = dict(avg_ms=pg['total_ms'] / max(pg['launches'], 1), launches_per_step=pg['launches'] / n_prof,
                              ms_per_step=pg['total_ms'] / n_prof)
        ctx.prof_enable(False)
        dom = max((pass_name, hvp_name), key=lambda k: kern[k]['ms_per_step'])
        traffic = measured_traffic(dom) if args.config == 3 else None   # the committed PMC passes are of config 3
        # what the launches of a step actually execute (the formula above counts passes the schedule folds away: the grad-KL
        # backward rides the second-order pass's reverse sweep, the first epoch takes its inner pass from _adapt)
        exec_flops = sum(k['launches_per_step'] * k['flop_per_row'] * k['rows_per_launch'] for k in kern.values() if 'flop_per_row' in k) * world
        step_s = elapsed / args.steps
        for name, k in kern.items():
            t = measured_traffic(name) if args.config == 3 else None
            if t and t.get('mfma_busy_frac') is not None:
                k['mfma_busy_frac'] = t['mfma_busy_frac']          # committed PMC pass (sha-guarded like `traffic`)
        out['roofline'] = {'bound': 'mfma', 'kernel': dom, 'achieved': kern[dom]['tflops'], 'peak': FP32_PEAK_TFLOPS,
                           'unit': 'TFLOP/s', 'frac': kern[dom]['tflops'] / FP32_PEAK_TFLOPS,
                           'peak_note': 'algorithmic float32 FLOPs (SURVEY 8d: matmul only, 2 / MAC) over the FP32 peak of 157.3 TFLOP/s, as in '
                                        'every round (the arithmetic type is float32); the kernel issues them as float32-equivalent FP16 '
                                        'products (3 matrix instructions per float32 product, see config.numerics), so 157.3 is not its '
                                        'ceiling -- `pipe` prices the same launch against the matrix pipe it occupies',
                           'pipe': {'dtype': 'f16 x 3 (two-term split: hi.hi + hi.lo + lo.hi, float32 accumulation)',
                                    'executed_tflops': SPLIT_PRODUCTS * kern[dom]['tflops'], 'peak': F16_PEAK_TFLOPS,
                                    'frac': SPLIT_PRODUCTS * kern[dom]['tflops'] / F16_PEAK_TFLOPS,
                                    'float32_equivalent_peak': F16_PEAK_TFLOPS / SPLIT_PRODUCTS,
                                    'note': 'matrix FLOPs the split issues for the formula FLOPs (x3; the few K = act_dim products on '
                                            'the exact-FP32 instruction counted alike) over the dense FP16 peak: what is not matrix time is '
                                            'the split itself (2 vector instructions per value), tanh, the distribution epilogue and '
                                            'the waits of one wave per SIMD'},
                           'traffic': traffic['bytes'] if traffic else None, 'traffic_source': traffic['source'] if traffic else None,
                           'algorithmic_bytes_per_launch': 4 * (O + 2 * A + 2) * kern[dom]['rows_per_launch'],
                           'avg_launch_ms': kern[dom]['avg_ms'], 'kernels': kern,
                           'executed': {'flops_per_step': exec_flops, 'tflops_per_gpu': exec_flops / step_s / 1e12 / world,
                                        'frac': exec_flops / step_s / 1e12 / world / FP32_PEAK_TFLOPS,
                                        'note': 'sum over the pass launches of one step of (launches x SURVEY 8d FLOPs per row x rows), over the '
                                                'timed step: machine utilisation on the matmul work the step executes; end_to_end_tflops_per_gpu '
                                                'is the same time against the FORMULA FLOPs of the reference\'s graph'},
                           'end_to_end_tflops_per_gpu': None if trpo else value * ((fl['fwd_bwd'] + E * (2 * fl['fwd'] + 3 * fl['bwd'] + fl['hvp'])
                                                                                           + 2 * fl['fwd'] + fl['bwd']) / 2.0) / 1e12 / world}

    # ---- Stage A alone (the scan / fit / normalise chain against HBM): process_samples on the stream, nothing else ----
    if not args.no_roofline and rank == 0 and world == 1:
        n_a = 50
        ctx.switch_to_pre_update()
        for _ in range(5):
            ctx.process_samples(0, **opts)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(n_a):
            ctx.process_samples(0, **opts)
        ctx.sync()
        ms_a = 1e3 * (time.perf_counter() - t0) / n_a
        bytes_a = (4 * (O + 1) + 8) * M * N
        out['roofline']['stage_a'] = {'ms': ms_a, 'rows': M * N, 'algorithmic_bytes': bytes_a, 'GB_per_s': bytes_a / (ms_a * 1e-3) / 1e9,
                                      'peak_GB_per_s': HBM_PEAK_GBS, 'frac': bytes_a / (ms_a * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      'note': 'process_samples of one sampling step, back to back on the stream (returns scan, feature Gram '
                                              'matrix on FP64 MFMA, Cholesky fit, GAE scan, normalisation): %d B per row of algorithmic '
                                              'traffic (SURVEY 8d); latency-bound -- five dependent launches over %d rows' % (4 * (O + 1) + 8, M * N)}

    # ---- the same step through the plugin classes (what a user who swaps the imports runs: meta_trainer.py:96-142) ----
    if rank == 0 and world == 1 and not trpo and not args.no_plugin_path and not args.staged_only:
        try:
            out['plugin_path'] = plugin_path(cfg, theta0, E, eta, opts, value, out.get('h2d', {}).get('upload_ms_per_step'))
        except Exception as e:          # the secondary measurement must never cost the primary line
            print('bench.py: plugin-path measurement skipped: %r' % (e,), file=sys.stderr)

    # ---- CPU baseline: the C + OpenMP restatement ("port") on the host cores, rank 0, bounded sample ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(cfg, theta0, alpha, eta, opts, E, trpo=trpo, steps=3 if args.config != 4 else 1,
                                           oracle_tasks=8 if args.config != 4 else 2)

    ctx.close()
    # ---- N > 1: also time the other scaling mode on the same ranks (weak: a 40-task batch per GPU) ----
    if world > 1:
        other = 'weak' if args.scaling == 'strong' else 'strong'
        try:
            M_other = cfg['M'] * (world if other == 'weak' else 1)
            ctx2, it2, M2 = setup(M_other)
            el2, _ = run_timed(ctx2, it2, args.warmup, args.steps)
            out['weak_batch' if other == 'weak' else 'fixed_batch'] = {
                'meta_batch_size': M_other, 'tasks_per_gpu': M2, 'scaling': other,
                'value': M_other * N * (K + 1) * args.steps / el2, 'ms_per_step': 1e3 * el2 / args.steps}
            ctx2.close()
        except Exception as e:      # the secondary measurement must never cost the primary line
            print('bench.py: %s-scaling measurement skipped: %r' % (other, e), file=sys.stderr)
    if rank == 0:
        print(json.dumps(out))


def plugin_path(cfg, theta0, E, eta, opts, device_value, upload_ms):
    """Trainer.train()'s timed stages through the plugin classes, on the benchmark's own batch:
        MetaSampleProcessor.process_samples (step 0) -> ProMP._adapt -> process_samples (step 1) -> ProMP.optimize_policy
    (a) host_paths: the path dicts MetaSampler.obtain_samples returns, in host memory -- every step flattens and uploads them
        (PCIe, pageable), downloads returns / advantages, builds the per-task dicts;
    (b) device_resident: the same batch as a device sampler leaves it (DeviceSlabSampler / DevicePointEnvSampler: slabs already
        in HBM, DevicePaths.device_ref) -- no upload, the downloads and the dicts remain.
    Both hand the reference's return values to the caller; neither is `value`."""
    from collections import OrderedDict
    from promp_amd import session as session_mod
    from promp_amd.baselines.linear_baseline import LinearFeatureBaseline
    from promp_amd.meta_algos.pro_mp import ProMP
    from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
    from promp_amd.samplers.device_point_sampler import DevicePaths
    from promp_amd.samplers.meta_sample_processor import MetaSampleProcessor
    from promp_amd.utils import logger as plog
    plog.configure(quiet=True)
    M, P, T, O, A, hidden = cfg['M'], cfg['P'], cfg['T'], cfg['O'], cfg['A'], cfg['hidden']
    keep = session_mod.current()
    policy = MetaGaussianMLPPolicy(name='bench-policy', obs_dim=O, action_dim=A, meta_batch_size=M, hidden_sizes=hidden,
                                   rank=0, world=1, device_id=0)
    policy.set_params(policy._unflatten(np.asarray(theta0, np.float32)))
    proc = MetaSampleProcessor(baseline=LinearFeatureBaseline(), discount=opts['discount'], gae_lambda=opts['gae_lambda'],
                               normalize_adv=opts['normalize_adv'])
    algo = ProMP(policy=policy, inner_lr=0.1, meta_batch_size=M, num_inner_grad_steps=1, learning_rate=1e-3, num_ppo_steps=E,
                 clip_eps=0.3, target_inner_step=0.01, init_inner_kl_penalty=float(eta[0]), adaptive_inner_kl_penalty=True)
    rng = np.random.RandomState(4321)
    p0 = synthetic.make_paths(rng, theta0, M, P, T, O, A, hidden)
    policy.switch_to_pre_update()
    sd0 = proc.process_samples(p0, log=False)
    algo._adapt(sd0)
    th1 = np.stack([np.concatenate([v.reshape(-1) for v in d.values()]) for d in policy.policies_params_vals]).astype(np.float32)
    p1 = synthetic.make_paths(rng, th1, M, P, T, O, A, hidden)

    def step(b0, b1):
        policy.switch_to_pre_update()                          # meta_trainer.py:85
        s0 = proc.process_samples(b0, log=False)               # :105
        algo._adapt(s0)                                        # :116
        s1 = proc.process_samples(b1, log=False)               # :105
        algo.optimize_policy([s0, s1], log=False)              # :128

    def timed(b0, b1, n):
        for _ in range(2):
            step(b0, b1)
        policy.session.ctx.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            step(b0, b1)
        policy.session.ctx.sync()
        return 1e3 * (time.perf_counter() - t0) / n

    n = 10
    ms_plain = timed(p0, p1, n)          # independent arrays per path: process_samples concatenates 800 paths x 5 arrays, twice
    # the batch as MetaSampler.obtain_samples returns it for a fixed-horizon environment (this config): path dicts whose device-bound
    # fields are views of flat page-locked arrays the sampler filled while it collected (samplers/meta_sampler.py: HostPaths)
    from copy import copy as _shallow
    from promp_amd.samplers.meta_sampler import slab_backed
    h0 = slab_backed(OrderedDict((i, [dict(p, agent_infos=dict(p['agent_infos'])) for p in pl]) for i, pl in p0.items()))
    h1 = slab_backed(OrderedDict((i, [dict(p, agent_infos=dict(p['agent_infos'])) for p in pl]) for i, pl in p1.items()))
    ms_host = timed(h0, h1, n)
    # (b) the batch as a device sampler leaves it: slot k holds step k, the path dicts are views of the downloaded slab
    sess = policy.session
    dev = []
    for slot, paths in ((0, p0), (1, p1)):
        fl = _lib.flatten_paths(paths)
        sess.upload_flat(slot, fl)
        dp = DevicePaths(paths)
        dp.device_ref = (sess.serial, sess.upload_serial[slot], slot)
        dp.flat = fl
        dev.append(dp)
    proc.lazy_host_arrays = True          # opt-in: per-row results cross PCIe on first use
    fetched = _lib.LazyResults.fetch_count
    ms_dev = timed(dev[0], dev[1], n)
    fetched = _lib.LazyResults.fetch_count - fetched
    proc.lazy_host_arrays = False
    ms_dev_eager = timed(dev[0], dev[1], n)
    env_steps = 2 * M * P * T
    res = {'steps': n,
           'host_paths': {'ms_per_step': ms_host, 'value': env_steps / (ms_host * 1e-3),
                          'pcie_upload_ms_per_step': upload_ms,
                          'pcie_share': (upload_ms / ms_host) if upload_ms else None,
                          'note': 'path dicts in host memory as MetaSampler.obtain_samples returns them for a fixed-horizon environment '
                                  '(HostPaths: the dicts\' observations / actions / rewards / agent_infos are views of the flat page-locked '
                                  'arrays the sampler wrote each finished episode into, once): promp_upload_step of those arrays + downloads + '
                                  'per-task dicts, twice per step',
                          'plain_dicts': {'ms_per_step': ms_plain, 'value': env_steps / (ms_plain * 1e-3),
                                          'note': 'the same batch as independent arrays per path (a custom sampler, early-terminating '
                                                  'environments): process_samples first concatenates 800 paths x 5 arrays per sampling step'}},
           'device_resident': {'ms_per_step': ms_dev, 'value': env_steps / (ms_dev * 1e-3),
                               'ratio_to_value': (env_steps / (ms_dev * 1e-3)) / device_value,
                               'per_row_downloads': fetched,
                               'eager_host_arrays': {'ms_per_step': ms_dev_eager, 'value': env_steps / (ms_dev_eager * 1e-3),
                                                     'note': 'SampleProcessor.lazy_host_arrays = False: per step 2 x (returns + raw '
                                                             'advantages float64, advantages float32) come back over PCIe inside '
                                                             'process_samples and 2 x %d path dicts receive their views' % (M * P)},
                               'note': 'DevicePaths with a valid device_ref (what DeviceSlabSampler / DevicePointEnvSampler return), '
                                       'SampleProcessor.lazy_host_arrays = True (opt-in): '
                                       'no upload; the per-row results (returns, advantages) are handed out as arrays that cross '
                                       'PCIe on first use -- the loop, like the reference trainer, never reads them '
                                       '(per_row_downloads counts the fetches that did happen); baseline coefficients and per-path '
                                       'sums come back every call'}}
    policy.session._drop()
    session_mod._current = keep
    return res


def shard_ideal(config, world):
    """What the committed single-GPU measurements say an N-rank run of the FIXED 40-task batch can reach: one rank's share
    (40 / N tasks, kernels only, no collective: bench.py --shard-of N on one GPU, profiles/rNN_shard_timings.txt) against the whole
    batch on one GPU (profiles/rNN_bench.json).  The ceiling is below N because a pass launch carries fixed cost (parameter
    staging, end-of-segment sums) that does not shrink with the shard: DESIGN.md section 7."""
    import glob
    import re
    if config not in (3, 4):
        return None
    try:
        suffix = '' if config == 3 else '_config4'
        sf = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_shard_timings%s.txt' % suffix)))[-1]
        bf = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_bench%s.json' % suffix)))[-1]
        single = json.load(open(bf))['ms_per_step']
        shard = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r'shard-of (\d+): \d+ tasks on this GPU, ([0-9.]+) ms/step', open(sf).read())}
        if world not in shard:
            return None
        return {'single_gpu_ms_per_step': single, 'shard_ms_per_step_kernels_only': shard[world],
                'ceiling_speedup_before_exchange_latency': single / shard[world],
                'ceilings': {str(n): single / v for n, v in sorted(shard.items())},
                'source': '%s, %s' % (os.path.relpath(bf, ROOT), os.path.relpath(sf, ROOT)),
                'note': 'strong scaling of the named 40-task batch: one MI355X finishes the whole update in %.2f ms, and a pass '
                        'launch keeps its fixed cost on a %d-task shard; N x is reachable in the weak sense only (weak_batch)' % (single, 40 // world)}
    except Exception:
        return None


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_hbm_traffic.json, written by tools/summarize_round.py); bench.py itself cannot read PMCs."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_hbm_traffic.json')))
    if not files:
        return None
    try:
        doc = json.load(open(files[-1]))
        # the file names the kernel sources it was measured on (tools/summarize_round.py): counters of other sources are not
        # this run's traffic and are not reported as such
        if doc.get('_kernel_sources_sha256') != kernel_sources_sha256():
            return dict(bytes=None, source='%s is of other kernel sources (stale): not reported' % os.path.relpath(files[-1], ROOT))
        t = doc.get(kernel)
        return dict(bytes=t['hbm_bytes_per_launch'], mfma_busy_frac=t.get('mfma_busy_frac'),
                    source=os.path.relpath(files[-1], ROOT)) if t else None
    except Exception:
        return None


def kernel_sources_sha256():
    """content hash of promp_amd/csrc (what the counters of a traffic file were measured on)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'promp_amd', 'csrc')
    for name in sorted(os.listdir(d)):
        h.update(name.encode())
        h.update(open(os.path.join(d, name), 'rb').read())
    return h.hexdigest()


def cpu_baseline(cfg, theta0, alpha, eta, opts, E, trpo=False, steps=3, oracle_tasks=8):
    """The hot path on the host cores: oracle/promp_cpu.c, a C + OpenMP restatement of the reference's path (one thread per
    task, float32 policy arithmetic like the TF graph, float64 sample processing like NumPy / SciPy; pinned against the NumPy
    oracle by tests/test_oracle_cpu_port.py), timed on the WHOLE named batch for `steps` steps.  The reference's own TF-1
    graph cannot run here (TensorFlow absent), hence "kind": "port".  The float64 NumPy oracle's rate on a sample of the
    tasks is reported beside it."""
    from oracle import cpu_port, policy as op, promp as pm, sample_processing as sp
    M, P, T, O, A, hidden = cfg['M'], cfg['P'], cfg['T'], cfg['O'], cfg['A'], cfg['hidden']
    port = cpu_port.CpuPort(M, P, T, O, A, hidden)
    rng = np.random.RandomState(99)

    def raw(theta_tasks):
        fl = _lib.flatten_paths(synthetic.make_paths(rng, theta_tasks, M, P, T, O, A, hidden))
        th = np.asarray(theta_tasks, np.float32)
        ls = np.tile(th[-A:], (M, 1)) if th.ndim == 1 else th[:, -A:].copy()
        return dict(obs=fl['obs'], rew=fl['rew'], act=fl['act'], old_mean=fl['old_mean'], old_log_std=ls)
    raw0 = raw(theta0)
    adv0, _ = port.process_samples(raw0['obs'], raw0['rew'], **opts)
    th1 = port.adapt(theta0, alpha, dict(raw0, adv=adv0), inner='loglik' if trpo else 'ratio')
    raw1 = raw(th1)

    def trpo_step(theta):
        from promp_amd.optimizers.conjugate_gradient_optimizer import ConjugateGradientOptimizer
        from promp_amd.utils import logger as plog
        plog.configure(quiet=True)
        s0 = dict(raw0, adv=port.process_samples(raw0['obs'], raw0['rew'], **opts)[0])
        port.adapt(theta, alpha, s0, inner='loglik')
        s1 = dict(raw1, adv=port.process_samples(raw1['obs'], raw1['rew'], **opts)[0])

        class Ev(object):
            th = np.array(theta, np.float32)
            def _e(self, outer, grad): return port.meta_grad(self.th, alpha, 0.0, 0.0, s0, s1, inner='loglik', outer=outer, want_grad=grad)
            def loss(self): return self._e('ratio', False)[1]['loss']
            def constraint_val(self): return self._e('ratio', False)[1]['outer_kl']
            def gradient(self): return self._e('ratio', True)[0]
            def constraint_gradient(self): return self._e('kl', True)[0]
            def get_theta(self): return self.th
            def set_theta(self, t): self.th = np.asarray(t, np.float32)
        cg = ConjugateGradientOptimizer()
        ev = Ev()
        cg.build_graph(ev, 0.01)
        cg.constraint_val(); cg.loss(); cg.optimize(); cg.loss(); cg.constraint_val()
        return ev.th

    theta = np.array(theta0, np.float32)
    t0 = time.perf_counter()
    for _ in range(steps):
        if trpo:
            theta = trpo_step(theta)
        else:
            theta, res = port.promp_step(theta, alpha, float(eta[0]), 0.3, 1e-3, E, raw0, raw1, opts)
    dt = time.perf_counter() - t0
    out = {'value': M * P * T * 2 * steps / dt, 'unit': 'env-steps/s', 'cores': min(port.threads(), M), 'kind': 'port',
           'sample': 'oracle/promp_cpu.c (C + OpenMP restatement, one thread per task: %d busy threads on %d host cores), the whole '
                     '%d-task batch, %d full steps, %.1f s' % (min(port.threads(), M), os.cpu_count() or 1, M, steps, dt)}
    if not trpo:
        # the same port on ONE thread, a sample of the tasks, one full step (BASELINE.md 3.2-3.3 asks for both figures)
        Ms = max(1, min(4, M))
        port1 = cpu_port.CpuPort(Ms, P, T, O, A, hidden)
        port1.set_threads(1)
        sub = lambda r: {k: (v[:Ms * P * T] if k != 'old_log_std' else v[:Ms]) for k, v in r.items()}
        r0s, r1s = sub(raw0), sub(raw1)
        t1 = time.perf_counter()
        port1.promp_step(np.array(theta0, np.float32), alpha, float(eta[0]), 0.3, 1e-3, E, r0s, r1s, opts)
        dt1 = time.perf_counter() - t1
        port1.set_threads(os.cpu_count() or 1)
        out['single_thread'] = {'value': Ms * P * T * 2 / dt1, 'unit': 'env-steps/s', 'cores': 1,
                                'sample': 'the same C port on one thread, one full step on %d of the %d tasks, %.1f s' % (Ms, M, dt1)}
    if not trpo and oracle_tasks:
        # the float64 NumPy oracle on a sample of the tasks, one full step
        Ms = min(oracle_tasks, M)
        spec = op.PolicySpec(O, A, hidden)
        p0 = synthetic.make_paths(rng, theta0, Ms, P, T, O, A, hidden)
        t64, a64, e64 = theta0.astype(np.float64), alpha.astype(np.float64), eta.astype(np.float64)
        t0 = time.perf_counter()
        s0, _, _ = sp.process_samples_meta(p0, baseline_kind=sp.BASELINE_LINEAR_FEATURE, **opts)
        ad = pm.adapt(spec, [t64] * Ms, s0, a64)
        t_a = time.perf_counter()
        p1 = synthetic.make_paths(rng, np.stack(ad).astype(np.float32), Ms, P, T, O, A, hidden)   # not timed
        t_b = time.perf_counter()
        s1, _, _ = sp.process_samples_meta(p1, baseline_kind=sp.BASELINE_LINEAR_FEATURE, **opts)
        pm.optimize_policy(spec, t64, [s0, s1], a64, e64, 0.3, pm.AdamState(spec.n_params), 1e-3, E)
        dt2 = (t_a - t0) + (time.perf_counter() - t_b)
        out['numpy_oracle'] = {'value': Ms * P * T * 2 / dt2, 'unit': 'env-steps/s',
                               'sample': 'float64 NumPy oracle, one full step on %d of the %d tasks, %.1f s' % (Ms, M, dt2)}
    return out


if __name__ == '__main__':
    main()
