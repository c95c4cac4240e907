"""The DiCE oracle (oracle/dice.py) against outputs of the reference's own DiceMetaSampleProcessor and against
torch.autograd on a transcription of DICEMAML's forward graph (tests/golden/dice_*.npz, oracle/gen_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import dice, policy as op, sample_processing as sp
from tests import helpers

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
KIND = dict(zero=sp.BASELINE_ZERO, linear_feature=sp.BASELINE_LINEAR_FEATURE, linear_time=sp.BASELINE_LINEAR_TIME)


@pytest.mark.parametrize('name', ['default', 'ragged', 'raw', 'positive'])
def test_dice_sample_processor_matches_reference_outputs(name):
    g = np.load(os.path.join(GOLDEN, 'dice_proc_%s.npz' % name))
    meta = json.loads(str(g['meta']))
    paths = helpers.dice_paths_from_golden(g)
    kw = meta['kwargs']
    for i, plist in enumerate(paths.values()):
        out = dice.process_samples_dice(plist, meta['max_path_length'], KIND[meta['baseline']], kw['discount'],
                                        kw.get('normalize_adv', True), kw.get('positive_adv', False))
        np.testing.assert_array_equal(out['mask'], g['mask'][i])
        np.testing.assert_allclose(out['adjusted_rewards'], g['adjusted_rewards'][i], rtol=1e-7, atol=1e-8)
        np.testing.assert_array_equal(out['rewards'], g['padded_rewards'][i])
        np.testing.assert_array_equal(out['observations'], g['padded_observations'][i])


@pytest.mark.parametrize('name', ['retbase', 'retbase_raw'])
def test_dice_sample_processor_with_return_baseline_matches_reference_outputs(name):
    """return_baseline given (dice_sample_processor.py:113-124, 196-238): GAE advantages beside the DiCE rewards"""
    g = np.load(os.path.join(GOLDEN, 'dice_proc_%s.npz' % name))
    meta = json.loads(str(g['meta']))
    kw = meta['kwargs']
    for i, plist in enumerate(helpers.dice_paths_from_golden(g).values()):
        out = dice.process_samples_dice(plist, meta['max_path_length'], KIND[meta['baseline']], kw['discount'],
                                        kw.get('normalize_adv', True), kw.get('positive_adv', False),
                                        return_baseline_kind=KIND[meta['return_baseline']], gae_lambda=kw.get('gae_lambda', 1.0))
        np.testing.assert_allclose(out['adjusted_rewards'], g['adjusted_rewards'][i], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(out['advantages'], g['advantages'][i], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('name', ['k1_ragged', 'k2_small'])
def test_vpg_dice_meta_gradient_matches_torch_autograd(name):
    g = np.load(os.path.join(GOLDEN, 'vpgdice_autograd_%s.npz' % name))
    c, theta, all_slabs = helpers.dice_case_from_golden(g)
    spec = op.PolicySpec(c['O'], c['A'], c['hidden'])
    r = dice.meta_objective_and_grad(spec, theta, all_slabs, np.full(spec.n_params, c['alpha']), outer='vpg')
    assert abs(r['loss'] - float(g['loss'])) < 1e-10
    np.testing.assert_allclose(r['grad'], g['grad'], rtol=1e-7, atol=1e-9 * np.abs(g['grad']).max())


@pytest.mark.parametrize('name', ['k1_small', 'k1_ragged', 'k2_small', 'k1_hc', 'k1_long'])
def test_dice_meta_gradient_matches_torch_autograd(name):
    g = np.load(os.path.join(GOLDEN, 'dice_autograd_%s.npz' % name))
    c, theta, all_slabs = helpers.dice_case_from_golden(g)
    spec = op.PolicySpec(c['O'], c['A'], c['hidden'])
    r = dice.meta_objective_and_grad(spec, theta, all_slabs, np.full(spec.n_params, c['alpha']))
    assert abs(r['loss'] - float(g['loss'])) < 1e-10
    np.testing.assert_allclose(r['grad'], g['grad'], rtol=1e-7, atol=1e-9 * np.abs(g['grad']).max())


def test_dice_hvp_matches_finite_differences():
    g = np.load(os.path.join(GOLDEN, 'dice_autograd_k1_ragged.npz'))
    c, theta, all_slabs = helpers.dice_case_from_golden(g)
    spec = op.PolicySpec(c['O'], c['A'], c['hidden'])
    slab = all_slabs[0][0]
    v = np.random.RandomState(0).randn(spec.n_params)
    eps = 1e-5
    gp = dice.loss_and_grad(spec, theta + eps * v, slab, True)['grad']
    gm = dice.loss_and_grad(spec, theta - eps * v, slab, True)['grad']
    # (the gradient above is the log-likelihood one with FIXED weights w; the coupled term is what the magic box adds when
    #  the objective itself is differentiated twice, so the finite difference is taken of the exact DiCE gradient instead)
    h_ll = (gp - gm) / (2 * eps)
    from oracle import promp as pm
    np.testing.assert_allclose(pm.hvp(spec, theta, slab, v, 'loglik', True), h_ll, rtol=1e-5, atol=1e-7)
    # symmetry of the full DiCE Hessian: u^T H v == v^T H u
    u = np.random.RandomState(1).randn(spec.n_params)
    a, b = u @ dice.hvp(spec, theta, slab, v, True), v @ dice.hvp(spec, theta, slab, u, True)
    assert abs(a - b) < 1e-9 * max(1.0, abs(a))
