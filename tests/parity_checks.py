"""Parity checks HIP path vs oracle, shared by the emulated (CPU, tiny) and the real (-m gpu) tests.
Every check goes through the C ABI (promp_amd._lib.Context) of the library it is handed.

Stated tolerances (float32 device arithmetic vs float64 oracle):
  returns      : rtol 1e-6  (FP64 scan on device, rounded once to float32)
  advantages   : rtol 1e-4, atol 1e-5  (after normalisation)
  coefficients : predictions compared, not raw coefficients (normal equations are ill-conditioned);
                 raw coefficients rtol 1e-6 on well-conditioned fixtures
  loss / KL    : rtol 1e-4
  gradients    : 1e-4 of the max-norm (meta-gradient: 1e-3 in BASELINE.md; we hold 1e-4)
"""
import os

import numpy as np

from oracle import policy as op
from oracle import promp as pm
from oracle import sample_processing as sp
from promp_amd import _lib
from tests import helpers

KIND = dict(zero=0, linear_feature=1, linear_time=2)


def rel_max(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / (np.max(np.abs(b)) + 1e-30))


def run_sample_processing(lib, paths, kwargs, baseline, hidden=(32, 32)):
    fl = _lib.flatten_paths(paths)
    M, O = len(paths), fl['obs'].shape[1]
    ctx = _lib.Context(M, O, 2, hidden, 1, max_rows=len(fl['rew']), max_paths=len(fl['path_row_offsets']) - 1, lib=lib)
    ctx.upload_step(0, fl['task_path_offsets'], fl['path_row_offsets'], fl['obs'], fl['rew'])
    ctx.process_samples(0, baseline_kind=KIND[baseline], **kwargs)
    out = ctx.download_processed(0)
    ctx.close()
    return out


def check_sample_processing_golden(lib, name):
    """HIP path vs the reference's own outputs (tests/golden/sample_proc_*.npz)."""
    meta, paths, g = helpers.load_sample_proc(name)
    out = run_sample_processing(lib, paths, meta['kwargs'], meta['baseline'])
    np.testing.assert_allclose(out['returns'], g['returns'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out['advantages'], g['advantages'], rtol=1e-4, atol=1e-5)
    if meta['baseline'] != 'zero' and meta['extras'].get('obs_dtype', 'float32') == 'float32':
        np.testing.assert_allclose(out['coeffs'], g['coeffs'], rtol=1e-5, atol=1e-7)
    st = meta['stats']
    assert np.all(np.isfinite(out['path_returns0'])) and np.all(np.isfinite(out['path_undiscounted']))
    np.testing.assert_allclose(np.mean(out['path_returns0']), st['AverageDiscountedReturn'], rtol=1e-5)
    np.testing.assert_allclose(np.mean(out['path_undiscounted']), st['AverageReturn'], rtol=1e-5)
    np.testing.assert_allclose(np.std(out['path_undiscounted']), st['StdReturn'], rtol=1e-4)
    np.testing.assert_allclose(np.max(out['path_undiscounted']), st['MaxReturn'], rtol=1e-5)
    np.testing.assert_allclose(np.min(out['path_undiscounted']), st['MinReturn'], rtol=1e-5)
    # adj_avg_rewards moments (meta_sample_processor.py:40-44) from the per-path sums
    n = len(g['rewards'])
    mu = np.sum(out['path_undiscounted']) / n
    sd = np.sqrt(max(np.sum(out['path_reward_sumsq']) / n - mu * mu, 0.0))
    np.testing.assert_allclose((g['rewards'] - mu) / (sd + 1e-8), g['adj_avg_rewards'], rtol=1e-4, atol=1e-5)


def check_sample_processing_oracle(lib, seed, M, P, T, O, ragged, kwargs, baseline='linear_feature'):
    from promp_amd import synthetic
    rng = np.random.RandomState(seed)
    theta = synthetic.init_theta(rng, O, (8, 8), 2)
    paths = synthetic.make_paths(rng, theta, M, P, T, O, 2, (8, 8), ragged=ragged)
    out = run_sample_processing(lib, paths, kwargs, baseline)
    ref, coeffs, stats = sp.process_samples_meta(paths, baseline_kind=KIND[baseline], **kwargs)
    np.testing.assert_allclose(out['returns'], np.concatenate([r['returns'] for r in ref]), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out['advantages'], np.concatenate([r['advantages'] for r in ref]), rtol=1e-4, atol=1e-5)
    return out


def check_layout_reuse(lib, seed, M=3, P=3, T=40, O=5, order=('X', 'X2', 'Y', 'X', 'Y', 'Y', 'X2')):
    """one context, one slab set, batches uploaded in the order X, X' (same offsets, other data), Y (other offsets, same
    capacity), X again: a set described by the offsets it already holds keeps its device-side tables (set_step_layout's
    early exit), any other description rebuilds them -- every result must equal a fresh context's on the same batch"""
    from promp_amd import synthetic
    kwargs = dict(discount=0.99, gae_lambda=0.97, normalize_adv=True, positive_adv=False)
    rng = np.random.RandomState(seed)
    theta = synthetic.init_theta(rng, O, (8, 8), 2)
    X = synthetic.make_paths(rng, theta, M, P, T, O, 2, (8, 8), ragged=True)
    X2 = {k: [dict(p, observations=p['observations'] + 0.5, rewards=p['rewards'] * 2.0 - 1.0) for p in v] for k, v in X.items()}
    Y = synthetic.make_paths(rng, theta, M, P, T, O, 2, (8, 8), ragged=True)
    fls = {name: _lib.flatten_paths(b) for name, b in (('X', X), ('X2', X2), ('Y', Y))}
    assert np.array_equal(fls['X']['path_row_offsets'], fls['X2']['path_row_offsets'])
    assert not np.array_equal(fls['X']['path_row_offsets'], fls['Y']['path_row_offsets'])
    cap_rows = max(len(f['rew']) for f in fls.values())
    ctx = _lib.Context(M, O, 2, (32, 32), 1, max_rows=cap_rows, max_paths=M * P, lib=lib)
    for name in order:
        fl, paths = fls[name], {'X': X, 'X2': X2, 'Y': Y}[name]
        ctx.upload_step(0, fl['task_path_offsets'], fl['path_row_offsets'], fl['obs'], fl['rew'])
        ctx.process_samples(0, baseline_kind=KIND['linear_feature'], **kwargs)
        got = ctx.download_processed(0)
        fresh = run_sample_processing(lib, paths, kwargs, 'linear_feature')
        for key in ('returns', 'advantages', 'coeffs', 'path_returns0'):
            np.testing.assert_array_equal(got[key], fresh[key], err_msg='%s after %s' % (key, name))
    ctx.close()


def check_float64_rewards(lib, seed, M=2, P=3, T=60, O=4):
    """rewards of large magnitude with small float64 structure (1e4 + N(0,1)): the reference scans the env's float64
    rewards, so returns must match the oracle to float64 accuracy and the raw GAE advantages (differences of large
    numbers) far better than a float32 round trip of the rewards would allow"""
    from promp_amd import synthetic
    rng = np.random.RandomState(seed)
    theta = synthetic.init_theta(rng, O, (8, 8), 2)
    paths = synthetic.make_paths(rng, theta, M, P, T, O, 2, (8, 8), ragged=True)
    for plist in paths.values():
        for p in plist:
            p['rewards'] = 1e4 + rng.randn(len(p['rewards']))          # float64
    kwargs = dict(discount=0.99, gae_lambda=0.95, normalize_adv=False, positive_adv=False)
    fl = _lib.flatten_paths(paths)
    assert fl['rew'].dtype == np.float64
    ctx = _lib.Context(M, O, 2, (32, 32), 1, max_rows=len(fl['rew']), max_paths=len(fl['path_row_offsets']) - 1, lib=lib)
    ctx.upload_step(0, fl['task_path_offsets'], fl['path_row_offsets'], fl['obs'], fl['rew'])
    ctx.process_samples(0, baseline_kind=KIND['linear_time'], **kwargs)
    ret64, adv64 = ctx.download_raw(0)
    ctx.close()
    ref, _, _ = sp.process_samples_meta(paths, baseline_kind=KIND['linear_time'], **kwargs)
    np.testing.assert_allclose(ret64, np.concatenate([r['returns'] for r in ref]), rtol=1e-12)
    # float32 rewards would be off by ~5e-4 each (ulp of 1e4 in float32 is ~1e-3), i.e. ~1e-2 after the GAE scan
    np.testing.assert_allclose(adv64, np.concatenate([r['advantages'] for r in ref]), rtol=0, atol=2e-5)


def check_fit_retry_on_rank_deficient_features(lib, seed, M=2, P=3, T=40, O=4):
    """LinearBaseline.fit's "NaN -> reg *= 10, at most 5 tries" (linear_baseline.py:68-77).  Two identical observation columns
    make Phi^T Phi exactly singular and reg = 1e-13 is below one ulp of its diagonal, so the first factorization attempts meet a
    zero / negative pivot (NaN) and only a larger reg goes through.  The reference's lstsq returns the minimum-norm solution
    instead; both must PREDICT the same baseline (the coefficients of the duplicated columns are not identified)."""
    from promp_amd import synthetic
    rng = np.random.RandomState(seed)
    theta = synthetic.init_theta(rng, O, (8, 8), 2)
    paths = synthetic.make_paths(rng, theta, M, P, T, O, 2, (8, 8))
    for plist in paths.values():
        for p in plist:
            p['observations'][:, 1] = p['observations'][:, 0]                 # duplicate column: rank-deficient features
            p['observations'][:, 2] = 0.0                                      # a zero column (and a zero square column)
    kwargs = dict(discount=0.99, gae_lambda=0.97, normalize_adv=False)
    fl = _lib.flatten_paths(paths)
    ctx = _lib.Context(M, O, 2, (32, 32), 1, max_rows=len(fl['rew']), max_paths=len(fl['path_row_offsets']) - 1, lib=lib)
    ctx.upload_step(0, fl['task_path_offsets'], fl['path_row_offsets'], fl['obs'], fl['rew'])
    ctx.process_samples(0, baseline_kind=KIND['linear_feature'], reg_coeff=1e-13, **kwargs)
    out = ctx.download_processed(0)
    ctx.close()
    assert np.all(np.isfinite(out['coeffs'])) and np.all(np.isfinite(out['advantages']))
    ref, coeffs, _ = sp.process_samples_meta(paths, baseline_kind=KIND['linear_feature'], reg_coeff=1e-13, **kwargs)
    adv_ref = np.concatenate([r['advantages'] for r in ref])
    np.testing.assert_allclose(out['advantages'], adv_ref, rtol=1e-3, atol=1e-3 * np.abs(adv_ref).max())


def check_fit_phases_equal_one_launch(lib, seed, M=3, P=6, T=100, O=376):
    """The baseline fit of wide feature matrices runs as one launch per phase (k_fitw_panel / k_fitw_update / k_fitw_back) from
    FITW_ML_MIN_D columns on; PROMP_FIT_ONE_LAUNCH=1 keeps k_fit_wide alone.  Same elimination order, same arithmetic per entry:
    the coefficients agree bit for bit."""
    import os
    from promp_amd import synthetic
    rng = np.random.RandomState(seed)
    theta = synthetic.init_theta(rng, O, (8, 8), 2)
    paths = synthetic.make_paths(rng, theta, M, P, T, O, 2, (8, 8))
    fl = _lib.flatten_paths(paths)
    kwargs = dict(discount=0.99, gae_lambda=0.97, normalize_adv=True)
    outs = []
    for one in ('0', '1'):
        old = os.environ.get('PROMP_FIT_ONE_LAUNCH')
        os.environ['PROMP_FIT_ONE_LAUNCH'] = one
        try:
            ctx = _lib.Context(M, O, 2, (32, 32), 1, max_rows=len(fl['rew']), max_paths=len(fl['path_row_offsets']) - 1, lib=lib)
        finally:
            if old is None: del os.environ['PROMP_FIT_ONE_LAUNCH']
            else: os.environ['PROMP_FIT_ONE_LAUNCH'] = old
        ctx.upload_step(0, fl['task_path_offsets'], fl['path_row_offsets'], fl['obs'], fl['rew'])
        ctx.process_samples(0, baseline_kind=KIND['linear_feature'], **kwargs)
        outs.append(ctx.download_processed(0))
        ctx.close()
    assert np.array_equal(outs[0]['coeffs'], outs[1]['coeffs'])
    assert np.array_equal(outs[0]['advantages'], outs[1]['advantages'])


def make_ctx(lib, M, O, A, hidden, K, all_paths, n_tasks_global=None, hidden_act='tanh', output_act=None):
    R = max(sum(len(p['rewards']) for pl in paths.values() for p in pl) for paths in all_paths)
    NPaths = max(sum(len(pl) for pl in paths.values()) for paths in all_paths)
    return _lib.Context(M, O, A, hidden, K, max_rows=R, max_paths=NPaths, lib=lib, n_tasks_global=n_tasks_global, hidden_act=hidden_act,
                        output_act=output_act)


def check_loss_grad(lib, seed, M, P, T, O, A, hidden, ragged=False, compact_log_std=False, low_log_std=False, min_std=1e-6,
                    hidden_act='tanh', output_act=None):
    """low_log_std: some log_std entries below log(min_std), i.e. the tf.maximum clip is active.  At the default min_std = 1e-6
    values are not comparable in float32 (see below); with a benign min_std (0.5) they are, and are compared."""
    theta, all_slabs, all_paths = helpers.make_promp_case(seed, M, P, T, O, A, hidden, 1, ragged=ragged,
                                                          low_log_std=low_log_std, min_std=min_std)
    spec = op.PolicySpec(O, A, hidden, min_std=min_std, hidden_act=hidden_act, output_act=output_act or 'identity')
    comparable = not low_log_std or min_std > 1e-3
    ctx = make_ctx(lib, M, O, A, hidden, 1, all_paths, hidden_act=hidden_act, output_act=output_act)
    ctx.set_min_std(min_std)
    helpers.upload_slabs(ctx, all_paths, all_slabs, compact_log_std=compact_log_std)
    rng = np.random.RandomState(seed + 1)
    th = (theta + (0.0 if low_log_std else 0.02) * rng.randn(M, theta.size)).astype(np.float32)
    ctx.set_task_thetas(th)
    for kind, name in ((0, 'ratio'), (1, 'clip'), (2, 'loglik')):
        for clip_ls in ((True,) if low_log_std else (False, True)):
            g, l, k = ctx.eval_loss_grad(1, kind, clip_eps=0.3, clip_log_std=clip_ls)
            for i in range(M):
                r = pm.loss_and_grad(spec, th[i].astype(np.float64), all_slabs[1][i], name, clip_ls, clip_eps=0.3)
                if comparable:
                    np.testing.assert_allclose(l[i], r['loss'], rtol=1e-4, atol=1e-6)
                    np.testing.assert_allclose(k[i], r['kl'], rtol=1e-4, atol=1e-6)
                    assert rel_max(g[i], r['grad']) < 1e-4, (name, i)
                else:
                    # sigma = 1e-6 makes z = (a-mu)/sigma amplify the float32 rounding of mu by 1e6, so values
                    # are not comparable in float32 (the oracle's mask arithmetic is pinned in float64 by
                    # tests/test_oracle_policy.py::k1_stdclip); here: finite results and an exact-zero mask
                    assert np.isfinite(l[i]) and np.isfinite(k[i]) and np.all(np.isfinite(g[i]))
                if low_log_std:   # gradient does not flow into clipped log_std entries
                    mask = th[i][-A:] < np.log(min_std)
                    assert mask.any() and np.all(g[i][-A:][mask] == 0.0)
    ctx.close()


def check_hvp(lib, seed, M, P, T, O, A, hidden, ragged=False, hidden_act='tanh', output_act=None):
    theta, all_slabs, all_paths = helpers.make_promp_case(seed, M, P, T, O, A, hidden, 1, ragged=ragged)
    spec = op.PolicySpec(O, A, hidden, hidden_act=hidden_act, output_act=output_act or 'identity')
    ctx = make_ctx(lib, M, O, A, hidden, 1, all_paths, hidden_act=hidden_act, output_act=output_act)
    helpers.upload_slabs(ctx, all_paths, all_slabs)
    rng = np.random.RandomState(seed + 2)
    th = (theta + 0.02 * rng.randn(M, theta.size)).astype(np.float32)
    ctx.set_task_thetas(th)
    v = rng.randn(M, theta.size).astype(np.float32)
    for kind, name in ((0, 'ratio'), (1, 'loglik')):
        out = ctx.eval_hvp(0, v, inner_kind=kind, clip_log_std=True, kl_weight=0.37)
        for i in range(M):
            t64 = th[i].astype(np.float64)
            ref = -pm.hvp(spec, t64, all_slabs[0][i], v[i].astype(np.float64), name, True) + \
                0.37 * pm.loss_and_grad(spec, t64, all_slabs[0][i], name, True)['grad_kl']
            assert rel_max(out[i], ref) < 1e-4, (name, i)
    ctx.close()


def check_split_accuracy(lib, hidden, O, A, seed=11, M=2, P=2, T=48, tol=2.5e-6, meta_tol=None):
    """Guard of the BF16-split GEMMs (k_pass: every GEMM; k_chain_hvp: layer 2): on a small well-conditioned case the gradient,
    the Hessian-vector product and the meta-gradient must agree with the float64 oracle to float32 rounding.  Measured on the
    MI355X (profiles/r03_split_accuracy.txt, tools/split_accuracy_gpu.py): six products of the 3-way split 1.7e-7 ... 1.0e-6 of
    the result's max-norm; the same kernels built with THREE products 5e-6 ... 5e-5.  tol = 2.5e-6 sits between the two: a
    regression of any GEMM to fewer products (or to plain BF16) fails here, long before the 1e-4 of the functional tests.
    128-wide layers (k_wb_fwd_bwd / k_wb_hvp, round 5; profiles/r05_split_accuracy.txt): gradient 3.4e-7 ... 1.7e-6, Hessian-vector
    product 5.3e-7 ... 1.2e-6 -- inside the same 2.5e-6 --, meta-gradient 2.7e-6 / 6.3e-6 (obs 20 / 111), where the exact-FP32
    cooperative kernels (PROMP_WIDE_FP32=1) measure 3.2e-6 / 3.8e-6 on the same cases: at K = 128 that is float32 accumulation,
    not the split, so the composite is held to meta_tol = 1e-5 there (three products would be ~5e-5)."""
    meta_tol = tol if meta_tol is None else meta_tol
    theta, all_slabs, all_paths = helpers.make_promp_case(seed, M, P, T, O, A, hidden, 1)
    spec = op.PolicySpec(O, A, hidden)
    ctx = make_ctx(lib, M, O, A, hidden, 1, all_paths)
    helpers.upload_slabs(ctx, all_paths, all_slabs)
    rng = np.random.RandomState(seed + 1)
    th = (theta + 0.02 * rng.randn(M, theta.size)).astype(np.float32)
    ctx.set_task_thetas(th)
    for kind, name in ((0, 'ratio'), (2, 'loglik')):
        g, _, _ = ctx.eval_loss_grad(1, kind, clip_eps=0.3, clip_log_std=False)
        for i in range(M):
            ref = pm.loss_and_grad(spec, th[i].astype(np.float64), all_slabs[1][i], name, False, clip_eps=0.3)['grad']
            assert rel_max(g[i], ref) < tol, ('gradient', name, i, rel_max(g[i], ref))
    v = rng.randn(M, theta.size).astype(np.float32)
    for kind, name in ((0, 'ratio'), (1, 'loglik')):
        hv = ctx.eval_hvp(0, v, inner_kind=kind, clip_log_std=True, kl_weight=0.37)
        for i in range(M):
            t64 = th[i].astype(np.float64)
            ref = -pm.hvp(spec, t64, all_slabs[0][i], v[i].astype(np.float64), name, True) + \
                0.37 * pm.loss_and_grad(spec, t64, all_slabs[0][i], name, True)['grad_kl']
            assert rel_max(hv[i], ref) < tol, ('hvp', name, i, rel_max(hv[i], ref))
    alpha, eta = np.full(spec.n_params, 0.1, np.float32), np.array([5e-4], np.float32)
    ctx.set_theta(theta)
    ctx.set_step_sizes(alpha)
    r = pm.meta_objective_and_grad(spec, theta.astype(np.float64), all_slabs, alpha.astype(np.float64), eta.astype(np.float64), 0.3)
    for cache in (0, 1):          # the recomputing and the cached second-order pass
        ctx.set_primal_cache(cache)
        g, _ = ctx.meta_grad(0.3, eta)
        assert rel_max(g, r['grad']) < meta_tol, ('meta-gradient', cache, rel_max(g, r['grad']))
    ctx.close()


def check_split_range(lib, hidden, O, A, seed=17, M=2, P=2, T=48, tol=2.5e-6, expect_redo=True, tail_from=7 / 16):
    """The FP16 split of the fused kernels has a range; the kernels keep every split operand inside it with exact powers of two that
    follow the data (promp_device.h: split_pair; promp_kernels_pass.h: pass_cotangent_scale, k_obs_range).  Float32 has no such
    range, so the results must not depend on any of this:
      * advantages spanning 12 orders of magnitude with the LARGE ones behind each wave's first tile (the cotangents leave the
        format at the scale the first tile set: the segment is walked again -- promp_split_events counts it),
      * all-zero advantages in the leading rows (no cotangent to set a scale from),
      * observations of size 3e3 and 3e-4, the hidden_0 kernel scaled the other way so that the network computes the same function
        (per-task power of two on the observations, its inverse on the kernel, taken off the kernel's gradient again),
      * a direction of size 1e-9 and 1e6 in the R-operator pass.
    Each against the float64 oracle at the accuracy guard's tolerance."""
    theta, all_slabs, all_paths = helpers.make_promp_case(seed, M, P, T, O, A, hidden, 1)
    spec = op.PolicySpec(O, A, hidden)
    rng = np.random.RandomState(seed + 1)
    th = (theta + 0.02 * rng.randn(M, theta.size)).astype(np.float32)
    n = [len(s['advantages']) for s in all_slabs[1]]
    cases = {}
    adv = [rng.randn(k).astype(np.float32) * np.float32(1e-7) for k in n]
    for a in adv:
        a[int(len(a) * tail_from):] *= np.float32(1e12)    # 1e-7 in the leading tiles / rounds, 1e5 later
    cases['heavy_tail'] = (adv, 1.0)
    adv = [rng.randn(k).astype(np.float32) for k in n]
    for a in adv:
        a[:min(len(a) - 3, 70)] = 0.0                   # the first tiles of every wave have no cotangent at all
    cases['leading_zeros'] = (adv, 1.0)
    cases['big_obs'] = ([s['advantages'] for s in all_slabs[1]], 3e3)
    cases['small_obs'] = ([s['advantages'] for s in all_slabs[1]], 3e-4)
    redone = 0
    for name, (adv, oscale) in cases.items():
        slabs = [[dict(s, observations=(s['observations'] * np.float32(oscale)).astype(np.float32)) for s in step] for step in all_slabs]
        for i in range(M):
            slabs[1][i] = dict(slabs[1][i], advantages=adv[i])
        paths = []
        for step in all_paths:
            q = type(step)()
            for key, plist in step.items():
                q[key] = [dict(p, observations=(p['observations'] * np.float32(oscale)).astype(np.float32)) for p in plist]
            paths.append(q)
        thc = th.copy()
        thc[:, :O * hidden[0]] *= np.float32(1.0 / oscale)
        ctx = make_ctx(lib, M, O, A, hidden, 1, paths)
        helpers.upload_slabs(ctx, paths, slabs)
        ctx.set_task_thetas(thc)
        ctx.split_events()
        for kind, oname in ((0, 'ratio'), (2, 'loglik')):
            g, l, _ = ctx.eval_loss_grad(1, kind, clip_eps=0.3, clip_log_std=False)
            for i in range(M):
                r = pm.loss_and_grad(spec, thc[i].astype(np.float64), slabs[1][i], oname, False, clip_eps=0.3)
                assert np.all(np.isfinite(g[i])), (name, oname, i)
                assert rel_max(g[i], r['grad']) < tol, (name, oname, i, rel_max(g[i], r['grad']))
        ev = ctx.split_events()
        if name == 'heavy_tail':
            redone = ev['pass_segments']
        for vscale in (1.0, 1e-9, 1e6):
            v = (vscale * rng.randn(M, theta.size)).astype(np.float32)
            v[:, :O * hidden[0]] *= np.float32(1.0 / oscale)      # (a direction in the parameters' own units)
            hv = ctx.eval_hvp(1, v, inner_kind=0, clip_log_std=True, kl_weight=0.37)
            for i in range(M):
                t64 = thc[i].astype(np.float64)
                ref = -pm.hvp(spec, t64, slabs[1][i], v[i].astype(np.float64), 'ratio', True) + \
                    0.37 * pm.loss_and_grad(spec, t64, slabs[1][i], 'ratio', True)['grad_kl']
                assert np.all(np.isfinite(hv[i])), (name, vscale, i)
                assert rel_max(hv[i], ref) < tol, (name, 'hvp', vscale, i, rel_max(hv[i], ref))
        if name == 'heavy_tail':
            redone += ctx.split_events()['hvp_segments']
        ctx.close()
    if expect_redo:
        assert redone > 0, 'the heavy-tailed case was meant to walk a segment twice'


ADAM_GRAD_FLOOR = 1e-3      # entries whose |g| stays above this fraction of the gradient's max-norm in EVERY epoch are compared
ADAM_STEP_TOL = 5e-3        # ... to this fraction of ONE Adam step (lr) over the whole trajectory (measured on the MI355X: 5e-5 .. 2.5e-4)
ADAM_MAX_EXCLUDED = 0.25    # and they must be most of the vector


def assert_adam_trajectory(d_dev, d_ref, grad_min_abs, grad_max_norm, lr, epochs, m_dev=None, v_dev=None, m_ref=None, v_ref=None):
    """The parameters after E epochs of tf.train.AdamOptimizer (maml_first_order_optimizer.py:22-46,82-115), per element.

    Adam divides the first moment by the root of the second: an entry moves by ~lr per epoch whatever the size of its gradient,
    in the direction of its sign.  The device gradient differs from the float64 one by ~1e-6 of the gradient's max-norm
    (float32 arithmetic), so an entry whose |g| is itself of that order may take a different direction -- legitimately, and by
    up to lr per epoch.  Every entry whose |g| stays above ADAM_GRAD_FLOOR x max-norm in every epoch (relative gradient error
    <= ~1e-3) must agree to ADAM_STEP_TOL x lr over the whole trajectory; the entries below the floor are bounded by what Adam
    can move them at all (the bias-corrected step never exceeds lr times a small factor), and must be a minority."""
    d_dev, d_ref = np.asarray(d_dev, np.float64), np.asarray(d_ref, np.float64)
    floor = ADAM_GRAD_FLOOR * float(np.min(grad_max_norm))
    judged = np.asarray(grad_min_abs) >= floor
    assert np.mean(~judged) <= ADAM_MAX_EXCLUDED, 'gradient floor excludes %.1f %% of the entries' % (100 * np.mean(~judged))
    err = np.abs(d_dev - d_ref)
    assert err[judged].max() <= ADAM_STEP_TOL * lr, (err[judged].max() / lr, int(np.argmax(err * judged)))
    assert np.max(np.abs(d_dev)) <= 1.5 * lr * epochs and err[~judged].max(initial=0.0) <= 2.0 * lr * epochs
    if m_dev is not None:
        # the moments: m is linear in the gradients (error ~ gradient error), v quadratic
        gs = float(np.max(grad_max_norm))
        assert np.max(np.abs(np.asarray(m_dev, np.float64) - m_ref)) <= 1e-4 * gs
        assert np.max(np.abs(np.asarray(v_dev, np.float64) - v_ref)) <= 2e-4 * gs * gs
    return float(err[judged].max() / lr), float(np.mean(~judged))


def check_adam_golden(lib, name):
    """E Adam epochs at BASELINE config 3's / config 4's network shapes against the torch.autograd + transcribed tf.train.Adam
    trajectory (tests/golden/promp_adam_*.npz): parameters per element, both moments, the losses before and after."""
    c, theta, all_slabs, g = helpers.load_promp_adam(name)
    M, O, A, hidden, K = c['M'], c['O'], c['A'], tuple(c['hidden']), c['K']
    N = all_slabs[0][0]['observations'].shape[0]
    ctx = _lib.Context(M, O, A, hidden, K, max_rows=M * N, max_paths=M, lib=lib)
    for k in range(K + 1):
        cat = lambda f: np.concatenate([f(s) for s in all_slabs[k]])
        ctx.upload_step(k, np.arange(M + 1), np.arange(M + 1) * N, cat(lambda s: s['observations']), np.zeros(M * N),
                        cat(lambda s: s['actions']), cat(lambda s: s['agent_infos']['mean']), cat(lambda s: s['agent_infos']['log_std']))
        ctx.set_advantages(k, cat(lambda s: s['advantages']))
    ctx.set_theta(theta)
    ctx.set_step_sizes(np.full(ctx.n_params, c['alpha'], np.float32))
    res = ctx.optimize(c['epochs'], c['lr'], c['clip_eps'], np.array(c['eta'], np.float32))
    np.testing.assert_allclose(res['loss_before'], float(g['losses'][0]), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(res['loss_after'], float(g['loss_after']), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(res['inner_kl'], g['inner_kl_after'], rtol=2e-4)
    m, v, t = ctx.get_adam_state()
    assert t == c['epochs']
    worst, excluded = assert_adam_trajectory(ctx.get_theta().astype(np.float64) - theta.astype(np.float64),
                                             g['theta_after'] - theta.astype(np.float64), g['grad_min_abs'], g['grad_max_norm'],
                                             c['lr'], c['epochs'], m_dev=m, v_dev=v, m_ref=g['adam_m'].astype(np.float64),
                                             v_ref=g['adam_v'].astype(np.float64))
    ctx.close()
    return worst, excluded


def check_meta(lib, seed, M, P, T, O, A, hidden, K, ragged=False, epochs=2, compact_log_std=False, hidden_act='tanh', output_act=None):
    """meta-objective + exact gradient, _adapt, and E Adam epochs + compute_stats."""
    theta, all_slabs, all_paths = helpers.make_promp_case(seed, M, P, T, O, A, hidden, K, ragged=ragged)
    spec = op.PolicySpec(O, A, hidden, hidden_act=hidden_act, output_act=output_act or 'identity')
    ctx = make_ctx(lib, M, O, A, hidden, K, all_paths, hidden_act=hidden_act, output_act=output_act)
    helpers.upload_slabs(ctx, all_paths, all_slabs, compact_log_std=compact_log_std)
    alpha = np.full(spec.n_params, 0.1, np.float32)
    eta = np.array([5e-4, 1e-3, 2e-3][:K], np.float32)
    a64, e64, t64 = alpha.astype(np.float64), eta.astype(np.float64), theta.astype(np.float64)
    ctx.set_theta(theta)
    ctx.set_step_sizes(alpha)
    g, st = ctx.meta_grad(0.3, eta)
    r = pm.meta_objective_and_grad(spec, t64, all_slabs, a64, e64, 0.3)
    np.testing.assert_allclose(st['loss'], r['loss'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(st['inner_kl'], r['inner_kl'], rtol=1e-4)
    np.testing.assert_allclose(st['outer_kl'], r['outer_kl'], rtol=1e-4)
    assert rel_max(g, r['grad']) < 1e-4
    # MAMLAlgo._adapt
    ctx.switch_to_pre_update()
    ctx.inner_adapt(0)
    ad = pm.adapt(spec, [t64] * M, all_slabs[0], a64)
    assert rel_max(ctx.get_task_thetas() - theta, np.stack(ad) - t64) < 1e-4
    # ProMP.optimize_policy core
    res = ctx.optimize(epochs, 1e-3, 0.3, eta)
    th_ref, ref = pm.optimize_policy(spec, t64, all_slabs, a64, e64, 0.3, pm.AdamState(spec.n_params), 1e-3, epochs)
    np.testing.assert_allclose(res['loss_before'], ref['loss_before'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(res['loss_after'], ref['loss_after'], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(res['inner_kl'], ref['inner_kl'], rtol=2e-4)
    np.testing.assert_allclose(res['outer_kl'], ref['outer_kl'], rtol=2e-4)
    # Adam trajectory, element by element (see assert_adam_trajectory): the float64 oracle's per-epoch gradients give the floor
    th_e, st_e, gmin, gmax = t64.copy(), pm.AdamState(spec.n_params), None, []
    for _ in range(epochs):
        ge = pm.meta_objective_and_grad(spec, th_e, all_slabs, a64, e64, 0.3)['grad']
        gmin = np.abs(ge) if gmin is None else np.minimum(gmin, np.abs(ge))
        gmax.append(np.max(np.abs(ge)))
        th_e = pm.adam_step(th_e, ge, st_e, 1e-3)
    m, v, t = ctx.get_adam_state()
    assert t == epochs
    assert_adam_trajectory(ctx.get_theta().astype(np.float64) - t64, th_ref - t64, gmin, np.array(gmax), 1e-3, epochs,
                           m_dev=m, v_dev=v, m_ref=st_e.m, v_ref=st_e.v)
    ctx.close()
    return res


def check_primal_cache(lib, seed, M, P, T, O, A, hidden, K=1, ragged=True):
    """the second-order pass reading the gradient pass's activations / means back from the primal cache must agree with
    the recomputing path to float32 rounding (tile blocks of ragged tasks, partial last tiles, every adaptation step),
    and both with the oracle; everything that does not go through the cache must not change at all"""
    theta, all_slabs, all_paths = helpers.make_promp_case(seed, M, P, T, O, A, hidden, K, ragged=ragged)
    spec = op.PolicySpec(O, A, hidden)
    alpha = np.full(spec.n_params, 0.1, np.float32)
    eta = np.array([5e-4, 1e-3, 2e-3][:K], np.float32)
    out = []
    for on in (True, False):
        ctx = make_ctx(lib, M, O, A, hidden, K, all_paths)
        ctx.set_primal_cache(on)
        helpers.upload_slabs(ctx, all_paths, all_slabs)
        ctx.set_theta(theta)
        ctx.set_step_sizes(alpha)
        g, st = ctx.meta_grad(0.3, eta)
        g2, _ = ctx.meta_grad(0.3, eta)          # a second evaluation overwrites the cache blocks in place
        np.testing.assert_array_equal(g, g2)
        out.append((g, st))
        ctx.close()
    (g_on, st_on), (g_off, st_off) = out
    for key in ('loss', 'inner_kl', 'outer_kl'):
        np.testing.assert_array_equal(st_on[key], st_off[key])
    assert rel_max(g_on, g_off.astype(np.float64)) < 2e-6
    r = pm.meta_objective_and_grad(spec, theta.astype(np.float64), all_slabs, alpha.astype(np.float64), eta.astype(np.float64), 0.3)
    assert rel_max(g_on, r['grad']) < 1e-4
    assert rel_max(g_off, r['grad']) < 1e-4


def check_split_path_equals_fused(lib, seed, M, P, T, O, A, hidden, epochs=3, attach_comm=False, fixed_order=False):
    """the several-rank launch sequence (k_reduce_final -> [ncclAllReduce] -> k_mean_adam) on ONE rank must reproduce the
    fused single-rank launch (k_final_adam) bit for bit: same column sums in the same order, same Adam arithmetic"""
    theta, all_slabs, all_paths = helpers.make_promp_case(seed, M, P, T, O, A, hidden, 1, ragged=True)
    spec = op.PolicySpec(O, A, hidden)
    eta = np.array([5e-4], np.float32)
    out = []
    for split in (False, True):
        ctx = make_ctx(lib, M, O, A, hidden, 1, all_paths)
        helpers.upload_slabs(ctx, all_paths, all_slabs)
        ctx.set_theta(theta)
        ctx.set_step_sizes(np.full(spec.n_params, 0.1, np.float32))
        if split:
            if attach_comm:
                ctx.comm_init(0, 1, _lib.comm_unique_id(lib))      # a real (one-rank) RCCL communicator: the all-reduce is enqueued
                ctx.comm_fixed_order(fixed_order)                  # ... or ncclAllGather + k_sum_ranks
            ctx.comm_split_path(True)
        res = ctx.optimize(epochs, 1e-3, 0.3, eta)
        g, st = ctx.meta_grad(0.3, eta)
        m, v, t = ctx.get_adam_state()
        out.append((ctx.get_theta(), m, v, t, res, g, st, ctx.reduced_get()))
        ctx.close()
    a, b = out
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_array_equal(x, y)
    assert a[3] == b[3] == epochs
    assert a[4] == pytest_approx_dict(b[4])
    np.testing.assert_array_equal(a[5], b[5])
    np.testing.assert_array_equal(a[7], b[7])


def on_policy_case(seed, M, P, T, O, A, hidden, K, alpha, inner_kind, ragged=True):
    """make_promp_case with the LAST step's old distribution replaced by the adapted policy's own outputs (what sampling
    with the adapted parameters records, meta_trainer.py:97-116): the operating point of TRPO's constraint"""
    theta, all_slabs, all_paths = helpers.make_promp_case(seed, M, P, T, O, A, hidden, K, ragged=ragged)
    spec = op.PolicySpec(O, A, hidden)
    thetas = np.tile(theta.astype(np.float64), (M, 1))
    for k in range(K):
        thetas = pm.adapt(spec, thetas, all_slabs[k], alpha.astype(np.float64), kind=inner_kind)
    for i, (slab, plist) in enumerate(zip(all_slabs[K], all_paths[K].values())):
        mean, log_std = op.forward(spec, thetas[i], slab['observations'].astype(np.float64), False)[:2]
        slab['agent_infos'] = dict(mean=mean.astype(np.float32), log_std=np.tile(log_std.astype(np.float32), (len(mean), 1)))
        r = 0
        for p in plist:
            n = len(p['rewards'])
            p['agent_infos'] = dict(mean=slab['agent_infos']['mean'][r:r + n], log_std=slab['agent_infos']['log_std'][r:r + n])
            r += n
    return theta, all_slabs, all_paths


def check_cg_solve_on_device(lib, seed, M, P, T, O, A, hidden, K=1, inner='loglik', cg_iters=4, fd=True):
    """promp_cg_solve (ConjugateGradientOptimizer's loop with its products enqueued back to back on the device) against the same
    loop on the host over the library's own products: conjugate_gradient_optimizer.py:325-354 with the exact product (no
    finite-difference noise: the two differ by the rounding of their dot products only) and with the reference's symmetric
    finite differences (one product and the closing quadratic form, where the directions are bitwise the same on both sides:
    over several iterations the noise of eps = 1e-5 in float32 amplifies any rounding difference).  The parameters are back
    where they were, the evaluation after the solve equals the one before it."""
    from promp_amd.optimizers.conjugate_gradient_optimizer import conjugate_gradients
    spec = op.PolicySpec(O, A, hidden)
    alpha = np.full(spec.n_params, 0.05, np.float32)
    kind = dict(loglik=_lib.INNER_LOGLIK, ratio=_lib.INNER_RATIO)[inner]
    okind = dict(loglik=pm.INNER_LOGLIK, ratio=pm.INNER_RATIO)[inner]
    theta, all_slabs, all_paths = on_policy_case(seed, M, P, T, O, A, hidden, K, alpha, okind)
    ctx = make_ctx(lib, M, O, A, hidden, K, all_paths)
    helpers.upload_slabs(ctx, all_paths, all_slabs)
    ctx.set_theta(theta)
    ctx.set_step_sizes(alpha)
    eta = np.zeros(K, np.float32)
    b, st0 = ctx.meta_grad(0.0, eta, inner_kind=kind, outer_kind=_lib.OUTER_RATIO)      # the loss gradient: the right-hand side
    # exact products
    fresh = [False]

    def hx_exact(x):
        out = ctx.constraint_hvp(np.asarray(x, np.float32), inner_kind=kind, refresh_chain=not fresh[0])
        fresh[0] = True
        return out
    ref = conjugate_gradients(hx_exact, b, cg_iters=cg_iters)
    ref_q = float(ref.astype(np.float64).dot(hx_exact(ref).astype(np.float64)))
    x, q = ctx.cg_solve(b, cg_iters=cg_iters, hvp_mode=2, inner_kind=kind)
    assert rel_max(x, ref) < 2e-4, rel_max(x, ref)
    assert q > 0 and abs(q - ref_q) <= 2e-4 * abs(ref_q), (q, ref_q)
    assert np.array_equal(ctx.get_theta(), theta)
    # with a Tikhonov term
    x2, q2 = ctx.cg_solve(b, cg_iters=2, reg_coeff=0.5, hvp_mode=2, inner_kind=kind)
    fresh[0] = False
    ref2 = conjugate_gradients(lambda v: hx_exact(v) + np.float32(0.5) * v, b, cg_iters=2)
    assert rel_max(x2, ref2) < 2e-4, rel_max(x2, ref2)
    assert abs(q2 - float(ref2.astype(np.float64).dot((hx_exact(ref2) + np.float32(0.5) * ref2).astype(np.float64)))) <= 2e-4 * abs(q2)
    # zero iterations: x = 0
    x0, q0 = ctx.cg_solve(b, cg_iters=0, hvp_mode=2, inner_kind=kind)
    assert not x0.any() and q0 == 0.0
    if fd:
        # the reference's finite differences: ONE iteration is step * b with step = b.b / b.(H b), and H b is taken at the same
        # displaced parameters on both sides (theta + eps b in one rounding here, two in NumPy: compared loosely)
        eps = np.float32(1e-5)

        def grad_c(th):
            ctx.set_theta(th.astype(np.float32))
            return ctx.meta_grad(0.0, eta, inner_kind=kind, outer_kind=_lib.OUTER_KL)[0]
        for mode in (0, 1):
            x1, q1 = ctx.cg_solve(b, cg_iters=1, eps=float(eps), hvp_mode=mode, inner_kind=kind)
            assert np.array_equal(ctx.get_theta(), theta)
            t64 = theta.astype(np.float64)
            ahead = grad_c(np.float32(1) * (theta + eps * b))
            hb = (ahead - grad_c(theta - eps * b)) / (2 * eps) if mode == 0 else (ahead - grad_c(theta)) / eps
            ctx.set_theta(theta)
            step = float(b.astype(np.float64).dot(b)) / float(b.astype(np.float64).dot(hb))
            # (the quotient of two noisy sums: the displaced parameters differ in their last bit between the two forms)
            assert np.isfinite(x1).all() and rel_max(x1, step * b) < 0.2, rel_max(x1, step * b)
            assert np.isfinite(q1)
        del t64
    # the solve leaves nothing behind: the same evaluation as before it
    b2, st2 = ctx.meta_grad(0.0, eta, inner_kind=kind, outer_kind=_lib.OUTER_RATIO)
    assert np.array_equal(b, b2) and st2['loss'] == st0['loss']
    ctx.close()


def check_exact_constraint_hvp(lib, seed, M, P, T, O, A, hidden, K=1, inner='loglik', tol=1e-4):
    """promp_constraint_hvp (2K+1 R-operator passes, Gauss-Newton form through the adaptation) against the float64 central
    difference of the oracle's constraint gradient, at the operating point of TRPO (old distribution = adapted policy)"""
    from oracle import trpo as otr
    spec = op.PolicySpec(O, A, hidden)
    alpha = np.full(spec.n_params, 0.05, np.float32)
    kind = dict(loglik=_lib.INNER_LOGLIK, ratio=_lib.INNER_RATIO)[inner]
    okind = dict(loglik=pm.INNER_LOGLIK, ratio=pm.INNER_RATIO)[inner]
    theta, all_slabs, all_paths = on_policy_case(seed, M, P, T, O, A, hidden, K, alpha, okind)
    ctx = make_ctx(lib, M, O, A, hidden, K, all_paths)
    helpers.upload_slabs(ctx, all_paths, all_slabs)
    ctx.set_theta(theta)
    ctx.set_step_sizes(alpha)
    ctx.set_primal_cache(False)        # the recomputing passes first (the default is on since round 6): the cached ones are compared with them below
    rng = np.random.RandomState(seed + 1)
    g, st = ctx.meta_grad(0.0, np.zeros(K, np.float32), inner_kind=kind, outer_kind=_lib.OUTER_KL)
    assert abs(st['outer_kl']) < 1e-6 and np.abs(g).max() < 1e-4      # on-policy: KL and its gradient vanish
    for trial in range(2):
        x = rng.randn(spec.n_params).astype(np.float32)
        if trial == 1:
            x[-A:] = 0.0                                              # a direction that leaves log_std alone
        hv = ctx.constraint_hvp(x, inner_kind=kind, refresh_chain=(trial == 0))
        ref = otr.constraint_hvp_fd64(spec, theta, all_slabs, alpha.astype(np.float64), x, inner_kind=okind)
        assert rel_max(hv, ref) < tol, rel_max(hv, ref)
        assert float(np.dot(x, hv)) > 0                               # positive semi-definite (J^T H_KL J), strictly here
    # symmetry: <y, H x> == <x, H y>
    x, y = rng.randn(spec.n_params).astype(np.float32), rng.randn(spec.n_params).astype(np.float32)
    hx = ctx.constraint_hvp(x, inner_kind=kind, refresh_chain=False)
    a = float(np.dot(y, hx))
    b = float(np.dot(x, ctx.constraint_hvp(y, inner_kind=kind, refresh_chain=False)))
    assert abs(a - b) <= 1e-3 * max(abs(a), abs(b))
    if tuple(hidden) in ((32, 32), (64, 64), (32, 64), (64, 32)) and O <= 32:
        # primal caches: the chain-refreshing passes store, all 2K + 1 R-operator passes of every product read; equal to the
        # recomputing passes to rounding, and nothing stale survives new parameters / a refresh
        assert ctx.constraint_hvp_cached_passes() == 0               # (switched off above)
        ctx.set_primal_cache(True)
        h1 = ctx.constraint_hvp(x, inner_kind=kind, refresh_chain=True)
        h2 = ctx.constraint_hvp(x, inner_kind=kind, refresh_chain=False)
        assert ctx.constraint_hvp_cached_passes() == 2 * (2 * K + 1)
        assert rel_max(h1, hx) < 5e-6 and np.array_equal(h1, h2), rel_max(h1, hx)
        ctx.set_theta(theta * np.float32(1.01))                      # new parameters without a refresh: the caches must not be read
        n0 = ctx.constraint_hvp_cached_passes()
        ctx.constraint_hvp(x, inner_kind=kind, refresh_chain=False)
        assert ctx.constraint_hvp_cached_passes() == n0
    ctx.close()


def check_schedule_invariance(lib, seed, M, P, T, O, A, hidden, K=1, iters=3, epochs=2):
    """launch scheduling must not change a single bit: sample processing of steps >= 1 on the second stream vs all on one
    stream, and the Hessian-vector pass's in-launch task reduction vs the separate reduction kernel.  Several iterations of
    process_samples(0) -> adapt -> process_samples(1..K) -> optimize on resident samples, with nothing synchronising the host
    in between, so that an ordering hole between the two streams (iteration n+1's processing overwriting advantages that
    iteration n's passes still read) would show up as a difference"""
    theta, all_slabs, all_paths = helpers.make_promp_case(seed, M, P, T, O, A, hidden, K, ragged=True)
    spec = op.PolicySpec(O, A, hidden)
    eta = np.full(K, 5e-4, np.float32)
    kwargs = dict(discount=0.99, gae_lambda=0.97, normalize_adv=True, positive_adv=False)
    out = []
    for overlap, fuse_min in ((0, 0), (1, 0), (0, 10 ** 6), (1, 10 ** 6)):
        ctx = make_ctx(lib, M, O, A, hidden, K, all_paths)
        ctx.set_schedule(overlap, fuse_min)
        helpers.upload_slabs(ctx, all_paths, all_slabs)
        ctx.set_theta(theta)
        ctx.set_step_sizes(np.full(spec.n_params, 0.1, np.float32))
        for _ in range(iters):
            ctx.switch_to_pre_update()
            ctx.process_samples(0, baseline_kind=KIND['linear_feature'], **kwargs)
            for k in range(K):
                ctx.inner_adapt(k)
                ctx.process_samples(k + 1, baseline_kind=KIND['linear_feature'], **kwargs)
            res = ctx.optimize(epochs, 1e-3, 0.3, eta)
        g, st = ctx.meta_grad(0.3, eta)
        proc = [ctx.download_processed(k) for k in range(K + 1)]
        out.append((ctx.get_theta(), g, ctx.get_task_thetas(), res, st, proc))
        ctx.close()
    ref = out[0]
    for other in out[1:]:
        np.testing.assert_array_equal(ref[0], other[0])
        np.testing.assert_array_equal(ref[1], other[1])
        np.testing.assert_array_equal(ref[2], other[2])
        assert ref[3] == pytest_approx_dict(other[3])
        for a, b in zip(ref[5], other[5]):
            for key in a:
                np.testing.assert_array_equal(a[key], b[key])


def check_adapt_reuse(lib, seed, M, P, T, O, A, hidden, K=1, iters=3, epochs=2, light=False):
    """promp_set_reuse_adapt: the first epoch of an optimisation takes the inner pass promp_inner_adapt just ran (theta', inner
    scalars, primal cache) instead of repeating it.  Must not change a single bit, must actually skip (one pass per optimisation
    from the second iteration on AND in the first, where the smallest log_std entry is known from promp_set_theta), and must NOT
    skip when something the pass reads changed in between or a log_std entry sits below log(min_std)."""
    theta, all_slabs, all_paths = helpers.make_promp_case(seed, M, P, T, O, A, hidden, K, ragged=True)
    spec = op.PolicySpec(O, A, hidden)
    eta = np.full(K, 5e-4, np.float32)
    kwargs = dict(discount=0.99, gae_lambda=0.97, normalize_adv=True, positive_adv=False)

    def run(reuse, cache, disturb=None, min_std=None):
        ctx = make_ctx(lib, M, O, A, hidden, K, all_paths)
        ctx.set_reuse_adapt(reuse)
        ctx.set_primal_cache(cache)
        if min_std is not None:
            ctx.set_min_std(min_std)
        helpers.upload_slabs(ctx, all_paths, all_slabs)
        ctx.set_theta(theta)
        ctx.set_step_sizes(np.full(spec.n_params, 0.1, np.float32))
        for it in range(iters):
            ctx.switch_to_pre_update()
            ctx.process_samples(0, baseline_kind=KIND['linear_feature'], **kwargs)
            for k in range(K):
                ctx.inner_adapt(k)
                ctx.process_samples(k + 1, baseline_kind=KIND['linear_feature'], **kwargs)
            if disturb == 'advantages' and it == iters - 1:
                ctx.set_advantages(0, ctx.download_processed(0)['advantages'] * np.float32(1.5))
            if disturb == 'step_sizes' and it == iters - 1:
                ctx.set_step_sizes(np.full(spec.n_params, 0.05, np.float32))
            res = ctx.optimize(epochs, 1e-3, 0.3, eta)
        g, st = ctx.meta_grad(0.3, eta)
        out = (ctx.get_theta(), g, res, st, ctx.adapt_passes_skipped())
        ctx.close()
        return out

    # (light: the emulated run keeps the cache-filling variant and one kind of disturbance; the GPU test runs everything)
    for cache in ((True,) if light else (False, True)):
        off, on = run(False, cache), run(True, cache)
        np.testing.assert_array_equal(off[0], on[0])
        np.testing.assert_array_equal(off[1], on[1])
        assert off[2] == pytest_approx_dict(on[2]) and off[3] == pytest_approx_dict(on[3])
        assert off[4] == 0 and on[4] == iters, (off[4], on[4])          # one pass per optimisation (the trailing meta_grad follows an Adam step)
        for what in (('advantages',) if light else ('advantages', 'step_sizes')):      # something the pass read has changed: the last optimisation repeats it
            a, b = run(False, cache, what), run(True, cache, what)
            np.testing.assert_array_equal(a[0], b[0])
            assert b[4] == iters - 1, (what, b[4])
    # a log_std entry below log(min_std): the inner step uses the raw value, the meta-objective's first step the clipped one -- never skipped
    a, b = run(False, True, min_std=2.0), run(True, True, min_std=2.0)
    np.testing.assert_array_equal(a[0], b[0])
    assert b[4] == 0


def check_staged_upload(lib, seed, M, P, T, O, A, hidden, iters=4, epochs=2):
    """promp_stage_step / promp_commit_step: two different batches alternate, the next one staged from pinned host arrays
    while the current one is optimised, nothing synchronising the host in between; every iteration's outcome must equal,
    bit for bit, the plain sequence upload_step -> process -> adapt -> process -> optimize on the same batches"""
    spec = op.PolicySpec(O, A, hidden)
    cases = [helpers.make_promp_case(seed + b, M, P, T, O, A, hidden, 1, ragged=(b == 1)) for b in range(2)]
    theta = cases[0][0]
    flat = [[_lib.flatten_paths(paths) for paths in c[2]] for c in cases]
    eta = np.array([5e-4], np.float32)
    kwargs = dict(discount=0.99, gae_lambda=0.97, normalize_adv=True, positive_adv=False)
    cap = [c[2] for c in cases]
    big = max(range(2), key=lambda b: sum(len(f['obs']) for f in flat[b]))

    def run(staged):
        ctx = make_ctx(lib, M, O, A, hidden, 1, cap[big])
        ctx.set_theta(theta)
        ctx.set_step_sizes(np.full(spec.n_params, 0.1, np.float32))
        args = lambda f: (f['task_path_offsets'], f['path_row_offsets'], f['obs'], f['rew'], f['act'], f['old_mean'], f['old_log_std'])
        if staged:      # pinned copies of both batches (the DMA sources)
            pin = [[{k: None for k in f} for f in fb] for fb in flat]
            for b in range(2):
                for k in range(2):
                    for key in ('obs', 'rew', 'act', 'old_mean', 'old_log_std'):
                        src = np.ascontiguousarray(flat[b][k][key], dtype=np.float32)
                        dst = _lib.pinned_empty(lib, src.shape)
                        dst[...] = src
                        pin[b][k][key] = dst
                    pin[b][k]['task_path_offsets'] = flat[b][k]['task_path_offsets']
                    pin[b][k]['path_row_offsets'] = flat[b][k]['path_row_offsets']
            for k in range(2):
                ctx.stage_step(k, *args(pin[0][k]))
        out = []
        for it in range(iters):
            b = it % 2
            if staged:
                ctx.commit_step(0), ctx.commit_step(1)
            else:
                for k in range(2):
                    ctx.upload_step(k, *args(flat[b][k]))
            ctx.switch_to_pre_update()
            ctx.process_samples(0, baseline_kind=KIND['linear_feature'], **kwargs)
            ctx.inner_adapt(0)
            ctx.process_samples(1, baseline_kind=KIND['linear_feature'], **kwargs)
            if staged and it + 1 < iters:
                for k in range(2):
                    ctx.stage_step(k, *args(pin[1 - b][k]))
            res = ctx.optimize(epochs, 1e-3, 0.3, eta)
            out.append((ctx.get_theta(), res, ctx.download_processed(1)['advantages']))
        ctx.close()
        return out

    for (ta, ra, aa), (tb, rb, ab) in zip(run(False), run(True)):
        np.testing.assert_array_equal(ta, tb)
        assert ra == pytest_approx_dict(rb)
        np.testing.assert_array_equal(aa, ab)


def pytest_approx_dict(d):
    class _Eq(dict):
        def __eq__(self, other):
            return all(np.array_equal(np.asarray(self[k]), np.asarray(other[k])) for k in self) and set(self) == set(other)
    return _Eq(d)


def check_learn_std_false(lib, seed, M, P, T, O, A, hidden):
    """GaussianMLPPolicy(learn_std=False): log_std is neither adapted by the inner step nor updated by Adam; everything else
    equals the oracle run with zero inner step sizes on the log_std entries"""
    theta, all_slabs, all_paths = helpers.make_promp_case(seed, M, P, T, O, A, hidden, 1)
    spec = op.PolicySpec(O, A, hidden)
    ctx = make_ctx(lib, M, O, A, hidden, 1, all_paths)
    helpers.upload_slabs(ctx, all_paths, all_slabs)
    alpha = np.full(spec.n_params, 0.1, np.float32)
    eta = np.array([5e-4], np.float32)
    ctx.set_theta(theta)
    ctx.set_learn_std(False)
    ctx.set_step_sizes(alpha)                     # (order must not matter: the mask is re-applied)
    a64 = alpha.astype(np.float64)
    a64[-A:] = 0.0
    g, st = ctx.meta_grad(0.3, eta)
    r = pm.meta_objective_and_grad(spec, theta.astype(np.float64), all_slabs, a64, eta.astype(np.float64), 0.3)
    np.testing.assert_allclose(st['loss'], r['loss'], rtol=1e-4, atol=1e-6)
    assert rel_max(g, r['grad']) < 1e-4
    ctx.switch_to_pre_update()
    ctx.inner_adapt(0)
    th1 = ctx.get_task_thetas()
    np.testing.assert_array_equal(th1[:, -A:], np.tile(theta[-A:], (M, 1)))
    assert np.all(np.any(th1[:, :-A] != theta[:-A], axis=1))
    ctx.optimize(3, 1e-3, 0.3, eta)
    th = ctx.get_theta()
    np.testing.assert_array_equal(th[-A:], theta[-A:])
    assert np.all(th[:-A] != theta[:-A]) or np.mean(th[:-A] != theta[:-A]) > 0.99
    with __import__('pytest').raises(_lib.PrompError, match='learn_std'):
        ctx.set_learn_std(True)
    ctx.close()


def check_trpo(lib, seed, M, P, T, O, A, hidden, inner_type='log_likelihood', cg_iters=10, max_backtracks=15,
               exploration=False, oracle_step=True, hvp_approach='finite_difference', on_policy=False):
    """TRPOMAML.optimize_policy (row a15) through the plugin classes.

    The reference's Hessian-vector product is a finite difference with eps = 1e-5 on float32 parameters
    (conjugate_gradient_optimizer.py:59-89): its result carries percent-level rounding noise that CG amplifies, so the
    final step of a float32 run is not comparable digit by digit with a float64 run.  Checked instead:
      (1) the device ingredients, tightly: loss, constraint value, loss gradient, constraint gradient at theta;
      (2) the step itself, as properties: descent, constraint satisfied, direction correlated with the float64 one.
    (The host CG / line-search logic is checked exactly in tests/test_trpo_host_logic.py.)"""
    from oracle import trpo as otrpo
    from promp_amd import session
    from promp_amd.meta_algos.trpo_maml import TRPOMAML
    from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
    from promp_amd.utils import logger
    logger.configure(quiet=True)
    _lib.set_library_for_testing(lib)
    try:
        spec = op.PolicySpec(O, A, hidden)
        kind = 'loglik' if inner_type == 'log_likelihood' else 'ratio'
        alpha = np.full(spec.n_params, 0.1)
        if on_policy:        # the last step's old distribution IS the adapted policy: TRPO's operating point
            theta, all_slabs, all_paths = on_policy_case(seed, M, P, T, O, A, hidden, 1, alpha.astype(np.float32), kind, ragged=False)
        else:
            theta, all_slabs, all_paths = helpers.make_promp_case(seed, M, P, T, O, A, hidden, 1)
        policy = MetaGaussianMLPPolicy(name='p', obs_dim=O, action_dim=A, meta_batch_size=M, hidden_sizes=hidden)
        policy.set_params(spec.to_ordered_dict(theta))
        algo = TRPOMAML(policy=policy, step_size=0.01, inner_type=inner_type, inner_lr=0.1, meta_batch_size=M,
                        num_inner_grad_steps=1, exploration=exploration, hvp_approach=hvp_approach)
        algo.optimizer._cg_iters = cg_iters
        algo.optimizer._max_backtracks = max_backtracks      # (the emulator run shortens the line search)
        samples = [[dict(observations=s['observations'], actions=s['actions'], advantages=s['advantages'],
                         agent_infos=s['agent_infos']) for s in step] for step in all_slabs]
        coeffs = None
        if exploration:      # E-MAML: cross-task reward z-scores of the last sampling step (what process_samples attaches)
            erng = np.random.RandomState(seed + 5)
            for d in samples[-1]:
                d['adj_avg_rewards'] = erng.randn(len(d['advantages'])).astype(np.float32) + 0.5
            coeffs = np.array([np.mean(d['adj_avg_rewards']) for d in samples[-1]], np.float64)
            algo._explore_coeffs = coeffs
            algo._explore_adv0 = np.concatenate([np.asarray(d['advantages'], np.float32) for d in samples[0]])
        # (1) ingredients at theta
        for k, sd in enumerate(samples):
            algo._slot_of(sd, k)
        ev = algo.optimizer._ev
        t64 = theta.astype(np.float64)
        r_loss = pm.meta_objective_and_grad(spec, t64, all_slabs, alpha, np.zeros(1), 0.0, kind, 'ratio')
        r_kl = pm.meta_objective_and_grad(spec, t64, all_slabs, alpha, np.zeros(1), 0.0, kind, 'kl')
        if exploration:
            xv, xg = otrpo.exploration_term(spec, t64, all_slabs[0], coeffs)
            r_loss = dict(r_loss, loss=r_loss['loss'] + xv, grad=r_loss['grad'] + xg)
        np.testing.assert_allclose(ev.loss(), r_loss['loss'], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(ev.constraint_val(), r_loss['outer_kl'], rtol=1e-4, atol=1e-6 if on_policy else 1e-7)
        assert rel_max(ev.gradient(), r_loss['grad']) < 1e-4
        if on_policy:        # KL(old || adapted) == 0 and stationary here: only rounding noise left on either side
            assert np.abs(ev.constraint_gradient()).max() < 1e-4 and np.abs(r_kl['grad']).max() < 1e-4
        else:
            assert rel_max(ev.constraint_gradient(), r_kl['grad']) < 1e-4
        # the optimizer's back-to-back loss / constraint queries at one parameter vector share ONE device evaluation, and a new
        # parameter vector (or anything else the evaluation reads) ends the sharing
        ctx, calls = algo.session.ctx, []
        orig = ctx.optimize              # (a forward evaluation is promp_optimize with zero epochs: DeviceSession.meta_eval)
        ctx.optimize = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            ev._memo = None
            l0, k0 = ev.loss(), ev.constraint_val()
            assert len(calls) == 1 + (1 if exploration else 0), len(calls)      # (the exploration term replaces step 0's advantages for a moment)
            ev.set_theta(theta * np.float32(1.01))
            l1 = ev.loss()
            assert len(calls) == 2 + (1 if exploration else 0) and l1 != l0
            ev.set_theta(theta)
            assert ev.loss() == l0 and ev.constraint_val() == k0
        finally:
            ctx.optimize = orig
        # (2) the step
        algo.optimize_policy(samples, log=False)
        st, last = algo.last_stats, algo.optimizer.last
        ref = None
        if oracle_step:      # (skipped at full BASELINE size: ~70 float64 meta-gradient evaluations of the NumPy oracle)
            ref = otrpo.trpo_maml_step(spec, theta, all_slabs, alpha, inner_kind=kind, max_kl=0.01, cg_iters=cg_iters,
                                       explore_coeffs=coeffs)
            d, dr = last['descent_direction'].astype(np.float64), ref['descent_direction']
            cos = d.dot(dr) / (np.linalg.norm(d) * np.linalg.norm(dr))
            if hvp_approach == 'exact':     # no finite-difference noise: the float32 run tracks the float64 one closely
                assert cos > 0.995, cos
                np.testing.assert_allclose(last['initial_step_size'], ref['initial_step_size'], rtol=2e-2)
                assert not last['rejected']
            elif np.isfinite(last['initial_step_size']):      # a negative-curvature FD estimate rejects the step (as the reference)
                assert cos > 0.5, cos
        else:                # the direction is a descent direction of the loss (g^T d > 0 for the step theta - s d)
            g = r_loss['grad']
            if np.isfinite(last['initial_step_size']):
                assert g.dot(last['descent_direction'].astype(np.float64)) > 0
        if not last['rejected']:
            assert st['loss_after'] < st['loss_before'] and st['mean_kl'] <= 0.01 * 1.001
        else:
            np.testing.assert_allclose(st['loss_after'], st['loss_before'], rtol=1e-6)     # parameters restored
        return st, ref
    finally:
        _lib.set_library_for_testing(None)
        session._current = None


def upload_dice_slabs(ctx, all_slabs):
    """steps 0..K of a DiCE case (oracle/dice.py:to_slab form) -> device slabs with the DiCE rewards"""
    for k, step in enumerate(all_slabs):
        n_paths = [len(sl['path_row_offsets']) - 1 for sl in step]
        tpo = np.concatenate([[0], np.cumsum(n_paths)]).astype(np.int32)
        lens = np.concatenate([np.diff(sl['path_row_offsets']) for sl in step])
        pro = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        cat = lambda f: np.concatenate([np.asarray(f(sl), dtype=np.float32) for sl in step])
        ctx.upload_step(k, tpo, pro, cat(lambda sl: sl['observations']), np.zeros(pro[-1], np.float32), cat(lambda sl: sl['actions']),
                        cat(lambda sl: sl['agent_infos']['mean']), cat(lambda sl: sl['agent_infos']['log_std']))
        ctx.set_dice_rewards(k, cat(lambda sl: sl['dice_rw']))


def check_dice(lib, name, tol=1e-4):
    """DICE-MAML on the device (PROMP_INNER_DICE / PROMP_OUTER_LOGLIK) against the oracle and, through the committed fixture,
    against torch.autograd on the reference's forward graph: gradient weights, inner step, exact meta-gradient."""
    from oracle import dice
    g = np.load(os.path.join(helpers.GOLDEN, 'dice_autograd_%s.npz' % name))
    c, t64, all_slabs = helpers.dice_case_from_golden(g)
    spec = op.PolicySpec(c['O'], c['A'], c['hidden'])
    M, K = c['M'], c['K']
    R = max(sum(len(sl['dice_rw']) for sl in step) for step in all_slabs)
    NPaths = max(sum(len(sl['path_row_offsets']) - 1 for sl in step) for step in all_slabs)
    ctx = _lib.Context(M, c['O'], c['A'], c['hidden'], K, max_rows=R, max_paths=NPaths, lib=lib)
    upload_dice_slabs(ctx, all_slabs)
    theta = t64.astype(np.float32)
    alpha = np.full(spec.n_params, c['alpha'], np.float32)
    ctx.set_theta(theta)
    ctx.set_step_sizes(alpha)
    # inner step (log-likelihood gradient with the suffix sums the device derives from the per-row rewards)
    ctx.switch_to_pre_update()
    ctx.inner_adapt(0, _lib.INNER_DICE)
    ad = dice.adapt(spec, [t64] * M, all_slabs[0], alpha.astype(np.float64))
    assert rel_max(ctx.get_task_thetas() - theta, np.stack(ad) - t64) < tol
    # exact meta-gradient through the adaptation, incl. the path-coupled second-order term
    grad, _ = ctx.meta_grad(0.0, np.zeros(K, np.float32), _lib.INNER_DICE, _lib.OUTER_LOGLIK)
    r = dice.meta_objective_and_grad(spec, t64, all_slabs, alpha.astype(np.float64))
    assert rel_max(grad, r['grad']) < tol
    assert rel_max(grad, g['grad']) < tol                      # torch.autograd on the padded magic-box graph
    # without the coupling term the gradient is measurably different (the check above is not vacuous)
    grad_ll, _ = ctx.meta_grad(0.0, np.zeros(K, np.float32), _lib.INNER_LOGLIK, _lib.OUTER_LOGLIK)
    assert rel_max(grad_ll, r['grad']) > 10 * tol
    ctx.close()


def check_vpg_dice(lib, name, tol=1e-4):
    """VPG-DiCE-MAML on the device: DiCE inner steps (PROMP_INNER_DICE), log-likelihood x advantage outer objective
    (PROMP_OUTER_LOGLIK with the last step's weights set to the advantages) against the oracle and, through the committed
    fixture, against torch.autograd on a transcription of VPG_DICEMAML's graph (vpg_dice_maml.py:35-113)."""
    from oracle import dice
    g = np.load(os.path.join(helpers.GOLDEN, 'vpgdice_autograd_%s.npz' % name))
    c, t64, all_slabs = helpers.dice_case_from_golden(g)
    spec = op.PolicySpec(c['O'], c['A'], c['hidden'])
    M, K = c['M'], c['K']
    R = max(sum(len(sl['dice_rw']) for sl in step) for step in all_slabs)
    NPaths = max(sum(len(sl['path_row_offsets']) - 1 for sl in step) for step in all_slabs)
    ctx = _lib.Context(M, c['O'], c['A'], c['hidden'], K, max_rows=R, max_paths=NPaths, lib=lib)
    upload_dice_slabs(ctx, all_slabs)
    ctx.set_advantages(K, np.concatenate([sl['vpg_advantages'] for sl in all_slabs[K]]).astype(np.float32))
    alpha = np.full(spec.n_params, c['alpha'], np.float32)
    ctx.set_theta(t64.astype(np.float32))
    ctx.set_step_sizes(alpha)
    grad, st = ctx.meta_grad(0.0, np.zeros(K, np.float32), _lib.INNER_DICE, _lib.OUTER_LOGLIK)
    r = dice.meta_objective_and_grad(spec, t64, all_slabs, alpha.astype(np.float64), outer='vpg')
    np.testing.assert_allclose(st['loss'], r['loss'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(st['loss'], float(g['loss']), rtol=1e-4, atol=1e-6)
    assert rel_max(grad, r['grad']) < tol
    assert rel_max(grad, g['grad']) < tol                      # torch.autograd on the padded graph
    # the DiCE outer objective on the same samples gives a measurably different gradient (the check above is not vacuous)
    ctx.set_dice_rewards(K, np.concatenate([sl['dice_rw'] for sl in all_slabs[K]]).astype(np.float32))
    grad_dice, _ = ctx.meta_grad(0.0, np.zeros(K, np.float32), _lib.INNER_DICE, _lib.OUTER_LOGLIK)
    assert rel_max(grad_dice, r['grad']) > 10 * tol
    ctx.close()


def check_comm_info_and_exchange_timing(lib):
    """what bench.py's `rccl` block is built from: promp_comm_info (ncclCommCount / ncclCommUserRank of the attached communicator, the
    device's PCI bus id) and the PROMP_KERNEL_EXCHANGE profiling slot (HIP events around every exchange)"""
    ctx = _lib.Context(2, 5, 3, (32, 32), 1, max_rows=64, max_paths=4, lib=lib)
    info = ctx.comm_info()
    assert info['nranks'] == 1 and info['rank'] == 0 and len(info['pci_bus_id']) >= 7
    ctx.comm_init(0, 1, _lib.comm_unique_id(lib))
    info = ctx.comm_info()
    assert info['nranks'] == 1 and info['rank'] == 0 and not info['fixed_order']
    ctx.comm_fixed_order(True)
    assert ctx.comm_info()['fixed_order']
    ctx.close()
    # the exchange's events: E epochs + the statistics pass, each timed
    theta, all_slabs, all_paths = helpers.make_promp_case(91, 2, 2, 16, 5, 3, (32, 32), 1)
    ctx = make_ctx(lib, 2, 5, 3, (32, 32), 1, all_paths)
    helpers.upload_slabs(ctx, all_paths, all_slabs)
    ctx.set_theta(theta)
    ctx.set_step_sizes(np.full(ctx.n_params, 0.1, np.float32))
    ctx.comm_init(0, 1, _lib.comm_unique_id(lib))
    ctx.comm_split_path(True)              # one rank takes the fused launch otherwise: no exchange to time
    ctx.prof_enable(True)
    ctx.optimize(3, 1e-3, 0.3, np.array([5e-4], np.float32))
    ex = ctx.prof_read(_lib.KERNEL_EXCHANGE)
    ctx.prof_enable(False)
    assert ex['launches'] == 4 and ex['total_ms'] >= 0, ex
    ctx.close()
