"""Kernel-source verification without a GPU: the HIP kernels + host sequencing of promp_amd/csrc compiled
against the SIMT interpreter (tests/emu) and compared with the oracle at tiny shapes.  These are NOT the
parity tests proper (those are -m gpu in test_gpu_parity.py and run the hipcc-built library on an MI355X);
they exist so that indexing / MFMA fragment-layout / sequencing bugs are caught in the build container."""
import pytest

from tests import devlib, helpers, parity_checks as pc


@pytest.fixture(scope='module')
def lib():
    return devlib.emu_library()


@pytest.fixture
def two_cus(monkeypatch):
    """sequencing tests launch dozens of kernels; two emulated CUs (half the host threads per launch) keep them short"""
    monkeypatch.setenv('PROMP_EMU_CUS', '2')


@pytest.mark.parametrize('name', ['default', 'gae095', 'positive', 'ragged', 'clipped_obs', 'zero_base', 'time_base',
                                  'undiscounted', 'float64_obs'])
def test_sample_processing_vs_reference_outputs(lib, name):
    pc.check_sample_processing_golden(lib, name)


def test_sample_processing_long_ragged_paths(lib):
    # T > 64 exercises the chunk carry of the wave scans; O=20 -> 3 FP64-MFMA feature blocks
    pc.check_sample_processing_oracle(lib, 5, M=2, P=3, T=150, O=20, ragged=True,
                                      kwargs=dict(discount=0.99, gae_lambda=0.97, normalize_adv=True))


def test_sample_processing_more_than_seventeen_feature_blocks(lib, two_cus):
    # obs_dim 140 -> 285 columns = 18 blocks of 16 = 6 bands of three, 21 squares: k_gram_tiled in two slices (blockIdx.y)
    pc.check_sample_processing_oracle(lib, 5, M=2, P=3, T=40, O=140, ragged=True,
                                      kwargs=dict(discount=0.99, gae_lambda=0.97, normalize_adv=True))


def test_sample_processing_ant_width_tiled_gram(lib, two_cus):
    # obs_dim 111 -> 227 columns = 15 blocks = 5 bands: 15 squares on the 16 waves of ONE workgroup, host-balanced wave map,
    # double-buffered 32-row feature tile, partial last rounds (ragged paths)
    pc.check_sample_processing_oracle(lib, 6, M=2, P=3, T=40, O=111, ragged=True,
                                      kwargs=dict(discount=0.99, gae_lambda=0.97, normalize_adv=True))


def test_sample_processing_short_last_band_and_untiled_gram(lib, two_cus, monkeypatch):
    # obs_dim 100 -> 205 columns = 13 blocks: the fifth band holds one block (the padding columns are zero and never written out);
    # then the same batch on k_gram_wide (PROMP_GRAM_UNTILED=1, the A/B switch)
    pc.check_sample_processing_oracle(lib, 7, M=2, P=2, T=40, O=100, ragged=True,
                                      kwargs=dict(discount=0.99, gae_lambda=0.97, normalize_adv=True))
    monkeypatch.setenv('PROMP_GRAM_UNTILED', '1')
    pc.check_sample_processing_oracle(lib, 7, M=2, P=2, T=40, O=100, ragged=True,
                                      kwargs=dict(discount=0.99, gae_lambda=0.97, normalize_adv=True))


def test_loss_grad_h64(lib):
    pc.check_loss_grad(lib, 7, M=2, P=2, T=37, O=20, A=6, hidden=(64, 64))




def test_split_range_follows_the_data(lib, two_cus):
    # FP16 split (round 6): the scales of cotangents, observations and directions follow the data, a segment whose cotangents leave
    # the format is walked again -- the kernel sources' control flow, checked here at the functional tolerance
    pc.check_split_range(lib, (32, 32), 7, 3, M=2, P=1, T=140, tol=1e-4)


def test_split_range_follows_the_data_h128(lib, two_cus):
    # the same on the cooperative 128-wide kernels (k_wb_fwd_bwd / k_wb_hvp: workgroup-wide scales, the work item walked again)
    # (two work items: rows [0, 32) and [32, 80); the second one's first round is all small, its second all large)
    pc.check_split_range(lib, (128, 128), 20, 6, M=1, P=1, T=80, tol=1e-4, tail_from=0.8)


def test_loss_grad_workgroups_straddling_two_tasks(lib):
    # 3 tasks x ~12 tiles on the emulator's 4 CUs (32 waves): every task gets >= 8 waves, so the wave-granular split
    # puts two tasks into one workgroup (two parameter copies in LDS, two partial rows out)
    pc.check_loss_grad(lib, 41, M=3, P=2, T=100, O=20, A=6, hidden=(64, 64), ragged=True)


def test_loss_grad_short_last_workgroup(lib):
    # 2 tasks x 10 tiles on 32 wave slots: one wave per tile (20 waves), the third workgroup has 4 spare waves
    pc.check_loss_grad(lib, 43, M=2, P=1, T=160, O=6, A=2, hidden=(32, 32))


def test_meta_workgroups_straddling_two_tasks(lib):
    pc.check_meta(lib, 42, M=3, P=2, T=90, O=7, A=3, hidden=(32, 32), K=1, epochs=1, ragged=True)


def test_loss_grad_h32_compact_log_std(lib):
    pc.check_loss_grad(lib, 8, M=2, P=1, T=70, O=3, A=2, hidden=(32, 32), compact_log_std=True)


def test_loss_grad_clipped_log_std(lib):
    pc.check_loss_grad(lib, 9, M=2, P=2, T=20, O=4, A=3, hidden=(32, 32), low_log_std=True)


def test_unequal_hidden_widths(lib):
    pc.check_loss_grad(lib, 15, M=1, P=2, T=40, O=6, A=3, hidden=(32, 64))
    pc.check_hvp(lib, 16, M=1, P=2, T=40, O=6, A=3, hidden=(64, 32))


def test_generic_policy_shapes(lib, two_cus):
    # three hidden layers / one layer / observations and actions beyond the fused kernels' tiles: promp_kernels_generic.h
    pc.check_loss_grad(lib, 7, M=2, P=2, T=37, O=9, A=3, hidden=(16, 24, 16), ragged=True)
    pc.check_hvp(lib, 8, M=2, P=1, T=40, O=9, A=3, hidden=(16, 24, 16))
    pc.check_loss_grad(lib, 9, M=2, P=1, T=30, O=140, A=17, hidden=(72,), compact_log_std=True)
    pc.check_meta(lib, 10, M=2, P=1, T=30, O=6, A=9, hidden=(20, 20), K=1, epochs=1)


@pytest.mark.parametrize('act', ['relu', 'identity'])
def test_hidden_nonlinearities_other_than_tanh(lib, two_cus, act):
    # policies/networks/mlp.py:47 takes any hidden_nonlinearity; relu and None (= linear hidden layers) run on the layer-by-layer
    # kernels whatever the widths -- here a shape the fused tanh kernels would otherwise take
    pc.check_loss_grad(lib, 41, M=2, P=1, T=33, O=6, A=3, hidden=(32, 32), hidden_act=act)
    pc.check_hvp(lib, 42, M=1, P=1, T=40, O=6, A=3, hidden=(32, 32), hidden_act=act)
    pc.check_meta(lib, 43, M=2, P=1, T=30, O=6, A=3, hidden=(32, 32), K=1, epochs=1, hidden_act=act)


@pytest.mark.parametrize('act,out', [('tanh', 'tanh'), ('relu', 'relu')])
def test_output_nonlinearity(lib, two_cus, act, out):
    # policies/networks/mlp.py:53-60, 114-117: output_nonlinearity on the mean network's last layer (None in every run script of the
    # reference); any policy that has one runs on the layer-by-layer kernels: objective, gradient, R-operator product, Adam epochs
    pc.check_loss_grad(lib, 44, M=2, P=1, T=33, O=6, A=3, hidden=(32, 32), hidden_act=act, output_act=out)
    pc.check_hvp(lib, 45, M=1, P=1, T=40, O=6, A=3, hidden=(32, 32), hidden_act=act, output_act=out)
    pc.check_meta(lib, 46, M=2, P=1, T=30, O=6, A=3, hidden=(32, 32), K=1, epochs=1, hidden_act=act, output_act=out)


def test_hvp_segments_straddling_tasks(lib, monkeypatch):
    # 3 tasks x ~12 tiles on 2 emulated CUs: the round list is cut into two shares, so a workgroup of k_chain_hvp walks
    # segments of two (or all three) tasks one after the other and is the last arriver for some of them
    monkeypatch.setenv('PROMP_EMU_CUS', '2')
    pc.check_hvp(lib, 17, M=3, P=2, T=100, O=20, A=6, hidden=(64, 64), ragged=True)


def test_exact_constraint_hvp_through_the_adaptation(lib, two_cus):
    pc.check_exact_constraint_hvp(lib, 23, M=2, P=1, T=24, O=5, A=3, hidden=(32, 32), K=1)


def test_staged_upload_sequence(lib, two_cus):
    pc.check_staged_upload(lib, 29, M=2, P=1, T=20, O=5, A=3, hidden=(32, 32), iters=2, epochs=1)


def test_step_layout_reuse_and_rebuild(lib, two_cus):
    pc.check_layout_reuse(lib, 61, M=2, P=2, T=24, O=3, order=('X', 'X2', 'Y', 'X'))


def test_float64_rewards(lib, two_cus):
    pc.check_float64_rewards(lib, 31, M=2, P=2, T=30, O=3)


def test_fused_and_separate_task_reduction_agree(lib, two_cus):
    pc.check_schedule_invariance(lib, 19, M=2, P=1, T=30, O=5, A=3, hidden=(32, 32), K=1, iters=1, epochs=1)


def test_first_epoch_reuses_the_inner_adapt_pass(lib, two_cus):
    pc.check_adapt_reuse(lib, 63, M=2, P=1, T=20, O=5, A=3, hidden=(32, 32), iters=2, epochs=1, light=True)


def test_primal_cache_matches_recomputation(lib, two_cus):
    # ragged tasks (partial last tiles, cache blocks of consecutive tasks 16 spare rows apart), unequal widths, K = 2
    # (more shapes, incl. 64/64 and 32/32, in the GPU test of the same name)
    pc.check_primal_cache(lib, 51, M=2, P=1, T=37, O=7, A=3, hidden=(32, 64), K=2)


def test_split_path_equals_fused_launch(lib, two_cus):
    pc.check_split_path_equals_fused(lib, 18, M=2, P=1, T=30, O=5, A=3, hidden=(32, 32), epochs=2)
    # the host sequence with a communicator attached (the emulated build's one-rank RCCL shim): all-reduce, and the fixed-order
    # exchange (all-gather + k_sum_ranks)
    pc.check_split_path_equals_fused(lib, 19, M=2, P=1, T=30, O=5, A=3, hidden=(32, 32), epochs=2, attach_comm=True)
    pc.check_split_path_equals_fused(lib, 19, M=2, P=1, T=30, O=5, A=3, hidden=(32, 32), epochs=2, attach_comm=True, fixed_order=True)


def test_learn_std_false(lib, two_cus):
    pc.check_learn_std_false(lib, 19, M=2, P=1, T=30, O=5, A=3, hidden=(32, 32))


def test_loss_grad_clipped_log_std_values_at_benign_min_std(lib):
    pc.check_loss_grad(lib, 20, M=2, P=2, T=20, O=4, A=3, hidden=(32, 32), low_log_std=True, min_std=0.5)


def test_fit_retries_with_larger_reg_on_rank_deficient_features(lib):
    pc.check_fit_retry_on_rank_deficient_features(lib, 21)


def test_hvp_h64(lib):
    pc.check_hvp(lib, 10, M=2, P=2, T=37, O=20, A=6, hidden=(64, 64))


def test_hvp_h32_odd_obs(lib):
    pc.check_hvp(lib, 11, M=1, P=2, T=50, O=5, A=3, hidden=(32, 32), ragged=True)


def test_hidden_widths_between_the_instantiated_ones_run_zero_padded(lib, two_cus):
    # (24, 40) on the (32, 64) kernels: parameter vectors in the caller's layout on both sides of the ABI
    pc.check_meta(lib, 14, M=2, P=1, T=30, O=6, A=3, hidden=(24, 40), K=1, ragged=True, epochs=1)


def test_adam_epochs_vs_torch_autograd_golden_h64(lib, two_cus):
    pc.check_adam_golden(lib, 'hc')


def test_meta_k1_h64(lib, two_cus):
    pc.check_meta(lib, 12, M=2, P=1, T=40, O=20, A=6, hidden=(64, 64), K=1, ragged=True, epochs=1)


def test_meta_k2_h32(lib, two_cus):
    pc.check_meta(lib, 13, M=2, P=2, T=33, O=5, A=3, hidden=(32, 32), K=2, epochs=2, compact_log_std=True)


@pytest.mark.parametrize('name', ['k1_ragged', 'k2_small'])
def test_dice_maml_gradient(lib, two_cus, name):
    pc.check_dice(lib, name)


def test_vpg_dice_maml_gradient(lib, two_cus):
    pc.check_vpg_dice(lib, 'k1_ragged')


def test_loss_grad_h32_wide_obs(lib):
    # O=31, A=8, H=32: the compact (non hidden_1) part of the gradient is larger than the hidden_1 kernel
    pc.check_loss_grad(lib, 14, M=1, P=1, T=40, O=31, A=8, hidden=(32, 32))


def test_hvp_h32_wide_obs(lib):
    pc.check_hvp(lib, 15, M=1, P=1, T=40, O=31, A=8, hidden=(32, 32))


# ---- cooperative kernels for hidden 128 / obs_dim > 32 (promp_kernels_policy_wide.h) and > 80 baseline features ----
def test_wide_loss_grad_ant_h128(lib):
    pc.check_loss_grad(lib, 23, M=2, P=1, T=70, O=111, A=8, hidden=(128, 128))      # 64-row rounds: one full, one partial


def test_wide_hvp_ant_h128(lib):
    pc.check_hvp(lib, 24, M=1, P=1, T=45, O=111, A=8, hidden=(128, 128))            # 32-row rounds


def test_wide_bf16_edge_shapes(lib, two_cus):
    # k_wb_*: one action, a task of 33 rows (a full round + one row), obs_dim 64 (the second observation class starts here)
    pc.check_loss_grad(lib, 27, M=2, P=1, T=33, O=64, A=1, hidden=(128, 128))
    pc.check_hvp(lib, 28, M=1, P=1, T=33, O=64, A=1, hidden=(128, 128))


def test_wide_hvp_h128_narrow_obs(lib):
    pc.check_hvp(lib, 25, M=1, P=2, T=20, O=20, A=6, hidden=(128, 128), ragged=True)


def test_wide_exact_constraint_hvp_h128(lib):
    # the TRPO constraint product through the cooperative kernels (KL objective in k_wide_hvp)
    pc.check_exact_constraint_hvp(lib, 26, M=1, P=2, T=20, O=20, A=6, hidden=(128, 128), K=1)


def test_wide_loss_grad_h64_wide_obs(lib):
    pc.check_loss_grad(lib, 21, M=1, P=1, T=40, O=40, A=8, hidden=(64, 64))


def test_wide_hvp_h64_wide_obs(lib):
    pc.check_hvp(lib, 27, M=1, P=1, T=33, O=100, A=2, hidden=(64, 64))


def test_sample_processing_first_wide_obs_dim(lib):
    # obs_dim 33 is the first size routed to k_gram_wide / k_fit_wide (5 feature blocks, like the largest k_gram<NBLK>)
    pc.check_sample_processing_oracle(lib, 333, M=2, P=3, T=60, O=33, ragged=True,
                                      kwargs=dict(discount=0.99, gae_lambda=0.97, normalize_adv=True))


def test_sample_processing_wide_features(lib):
    pc.check_sample_processing_oracle(lib, 31, M=2, P=3, T=70, O=111, ragged=True,
                                      kwargs=dict(discount=0.99, gae_lambda=0.97, normalize_adv=True))


def test_kl_objective_gradient(lib):
    # LOSS_KL (3): mean KL(old||new) and its gradient, the building block of the TRPO constraint
    import numpy as np
    from oracle import policy as op, promp as pm
    from promp_amd import _lib
    theta, all_slabs, all_paths = helpers.make_promp_case(16, 2, 2, 20, 5, 3, (32, 32), 1)
    spec = op.PolicySpec(5, 3, (32, 32))
    ctx = pc.make_ctx(lib, 2, 5, 3, (32, 32), 1, all_paths)
    helpers.upload_slabs(ctx, all_paths, all_slabs)
    th = np.tile(theta, (2, 1))
    ctx.set_task_thetas(th)
    g, l, k = ctx.eval_loss_grad(1, _lib.LOSS_KL)
    for i in range(2):
        r = pm.loss_and_grad(spec, theta.astype(np.float64), all_slabs[1][i], 'kl', False)
        np.testing.assert_allclose(l[i], r['loss'], rtol=1e-4)
        np.testing.assert_allclose(k[i], r['kl'], rtol=1e-4)
        assert pc.rel_max(g[i], r['grad']) < 1e-4
    ctx.close()


def test_trpo_e_maml_exploration_term(lib, two_cus):
    pc.check_trpo(lib, 18, M=2, P=1, T=12, O=4, A=2, hidden=(32, 32), cg_iters=1, max_backtracks=1, exploration=True)


def test_cg_solve_on_device(lib, two_cus):
    pc.check_cg_solve_on_device(lib, 19, M=2, P=1, T=16, O=4, A=2, hidden=(32, 32), cg_iters=2)


def test_trpo_maml_step(lib, two_cus):
    pc.check_trpo(lib, 17, M=2, P=1, T=16, O=4, A=2, hidden=(32, 32), cg_iters=1, max_backtracks=2)


def test_comm_info_and_exchange_timing_on_a_one_rank_communicator(lib, two_cus):
    pc.check_comm_info_and_exchange_timing(lib)


def test_sample_processing_fit_with_one_launch_per_phase(lib, two_cus, monkeypatch):
    # obs_dim 200 -> 404 columns >= FITW_ML_MIN_D: k_fitw_panel / k_fitw_update / k_fitw_back (13 panels of 32 columns) and the
    # only_bad launch of k_fit_wide behind them, against the oracle; then the same batch through k_fit_wide alone
    kw = dict(discount=0.99, gae_lambda=0.97, normalize_adv=True)
    pc.check_sample_processing_oracle(lib, 5, M=2, P=3, T=40, O=200, ragged=True, kwargs=kw)
    monkeypatch.setenv('PROMP_FIT_ONE_LAUNCH', '1')
    pc.check_sample_processing_oracle(lib, 5, M=2, P=3, T=40, O=200, ragged=True, kwargs=kw)
