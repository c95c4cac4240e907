"""The C + OpenMP restatement bench.py times as its CPU baseline (oracle/promp_cpu.c) against the float64 NumPy oracle,
which the reference's own outputs pin (sample processing) or torch.autograd pins (meta-gradient)."""
import numpy as np
import pytest

from oracle import cpu_port, policy as op, promp as pm, sample_processing as sp
from promp_amd import _lib, synthetic
from tests import helpers


def _flat(slabs, paths):
    fl = _lib.flatten_paths(paths)
    return dict(obs=fl['obs'], rew=fl['rew'], act=fl['act'], old_mean=fl['old_mean'],
                old_log_std=np.stack([s['agent_infos']['log_std'][0] for s in slabs]).astype(np.float32),
                adv=np.concatenate([s['advantages'] for s in slabs]).astype(np.float32))


@pytest.mark.parametrize('O,A,hidden', [(5, 3, (32, 32)), (20, 6, (64, 64))])
def test_cpu_port_matches_numpy_oracle(O, A, hidden):
    M, P, T = 3, 4, 30
    theta, all_slabs, all_paths = helpers.make_promp_case(91, M, P, T, O, A, hidden, 1)
    spec = op.PolicySpec(O, A, hidden)
    port = cpu_port.CpuPort(M, P, T, O, A, hidden)
    assert port.threads() >= 1
    s0, s1 = _flat(all_slabs[0], all_paths[0]), _flat(all_slabs[1], all_paths[1])
    # sample processing (float64 in both)
    adv, ret = port.process_samples(s0['obs'], s0['rew'], discount=0.99, gae_lambda=0.97, normalize_adv=True)
    ref, _, _ = sp.process_samples_meta(all_paths[0], baseline_kind=sp.BASELINE_LINEAR_FEATURE, discount=0.99, gae_lambda=0.97,
                                        normalize_adv=True)
    np.testing.assert_allclose(ret, np.concatenate([r['returns'] for r in ref]), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(adv, np.concatenate([r['advantages'] for r in ref]), rtol=2e-4, atol=2e-5)
    # meta-objective and exact gradient, ProMP kinds and the TRPO-MAML kinds
    alpha = np.full(spec.n_params, 0.1)
    for inner, outer in (('ratio', 'clip'), ('loglik', 'ratio'), ('loglik', 'kl')):
        g, st = port.meta_grad(theta, alpha, 5e-4, 0.3, s0, s1, inner=inner, outer=outer)
        r = pm.meta_objective_and_grad(spec, theta.astype(np.float64), all_slabs, alpha, np.array([5e-4]), 0.3, inner, outer)
        np.testing.assert_allclose(st['loss'], r['loss'], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(st['inner_kl'], r['inner_kl'][0], rtol=2e-4)
        np.testing.assert_allclose(st['outer_kl'], r['outer_kl'], rtol=2e-4)
        assert np.max(np.abs(g - r['grad'])) < 2e-4 * np.max(np.abs(r['grad']))
    th1 = port.adapt(theta, alpha, s0)
    ad = pm.adapt(spec, [theta.astype(np.float64)] * M, all_slabs[0], alpha)
    assert np.max(np.abs(th1 - np.stack(ad))) < 1e-5
