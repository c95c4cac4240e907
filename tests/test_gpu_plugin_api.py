"""The plugin-API scenarios of test_plugin_api.py on the real MI355X (product library, no emulator)."""
import pytest

from promp_amd import _lib, session
from tests import test_plugin_api as scen

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def product_library():
    _lib.set_library_for_testing(None)      # default: promp_amd/libpromp_hip.so
    yield
    session._current = None


def test_meta_sample_processor_api():
    scen.run_processor_scenario()


def test_policy_algo_api_halfcheetah_shapes():
    scen.run_algo_scenario(M=8, P=5, T=200, O=20, A=6, hidden=(64, 64), K=1, epochs=5)


def test_policy_algo_api_config2_full_size():
    # BASELINE config 2 as named: 8 tasks x 20 paths x 200 steps, HalfCheetah shapes, 2x64 MLP
    scen.run_algo_scenario(M=8, P=20, T=200, O=20, A=6, hidden=(64, 64), K=1, epochs=5)


def test_policy_algo_api_two_inner_steps():
    scen.run_algo_scenario(M=3, P=4, T=50, O=5, A=3, hidden=(32, 32), K=2, epochs=3)


def test_policy_algo_api_hidden_sizes_100():
    # the reference's other common policy size (hidden_sizes=(100, 100)): runs zero-padded on the (128, 128) kernels
    scen.run_algo_scenario(M=4, P=4, T=60, O=20, A=6, hidden=(100, 100), K=1, epochs=3)


def test_trainer_end_to_end_point_env():
    scen.run_trainer_scenario(n_itr=3)


def test_device_rollout_point_env():
    scen.run_device_rollout_scenario(M=4, B=20, T=100, reward_type='sparse')          # BASELINE config 1 shapes, default reward
    scen.run_device_rollout_scenario(M=5, B=3, T=33, hidden=(64, 64), reward_type='dense')
    scen.run_device_rollout_scenario(M=3, B=5, T=40, reward_type='dense_squared')


def test_device_rollout_point_env_with_device_noise():
    scen.test_device_rollout_point_env_with_device_noise(None)


def test_policy_step_fills_the_slab():
    scen.run_policy_step_scenario()
    scen.run_policy_step_scenario(M=8, B=20, T=12, O=20, A=6, hidden=(64, 64))       # HalfCheetah shapes


def test_rollout_kernels_on_any_layer_table():
    scen.test_rollout_kernels_on_any_layer_table(None)
    scen.run_policy_step_scenario(M=4, B=20, T=6, O=376, A=17, hidden=(64, 64))        # Humanoid's dimensions
    scen.run_policy_step_scenario(M=3, B=5, T=6, O=30, A=6, hidden=(256, 256), hidden_act='relu')
    scen.run_device_rollout_scenario(M=4, B=20, T=100, hidden=(64, 64, 64), reward_type='sparse')


def test_trainer_with_device_rollouts():
    scen.run_trainer_scenario(n_itr=3, device_rollouts=True)


def test_get_actions_on_device():
    scen.run_get_actions_scenario(M=8, B=20, O=20, A=6, hidden=(64, 64))
    scen.run_get_actions_scenario(M=4, B=20, O=376, A=17, hidden=(64, 64, 64))      # Humanoid's dimensions, three hidden layers


@pytest.mark.parametrize('device_rollouts', [False, True], ids=['host_rollouts', 'device_rollouts'])
def test_promp_learns_on_point_env(tmp_path, device_rollouts):
    """End to end on a real environment: the reference's point-mass recipe
    (run_scripts/pro-mp_run_point_mass.py shapes) improves the post-adaptation return."""
    import csv
    import importlib.util
    import os
    from promp_amd.utils import logger
    spec = importlib.util.spec_from_file_location('run_point', os.path.join(scen.__file__.rsplit('/tests/', 1)[0], 'run_scripts', 'pro-mp_run_point_mass.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = dict(mod.DEFAULT, n_itr=40, meta_batch_size=8, rollouts_per_meta_task=10, max_path_length=25, seed=3, device_rollouts=device_rollouts,
               reward_type='dense', normalize_env=False)      # dense reward, bare action box: a learning signal within 40 iterations
    logger.configure(dir=str(tmp_path), quiet=True)
    mod.main(cfg)
    rows = list(csv.DictReader(open(os.path.join(str(tmp_path), 'progress.csv'))))
    assert len(rows) == 40
    post = [float(r['Step_1-AverageReturn']) for r in rows]
    first, last = sum(post[:5]) / 5, sum(post[-5:]) / 5
    assert last > first + 0.05 * abs(first), (first, last)       # returns are negative distances: closer to the goal
    logger.configure(quiet=True)


@pytest.mark.parametrize('exploration', [False, True], ids=['trpo_maml', 'e_maml'])
def test_trpo_maml_run_script_on_point_env(tmp_path, exploration):
    """run_scripts/maml_run_point_mass.py (the reference's maml_run_mujoco.py / e-maml_run_mujoco.py on its MuJoCo-free environment):
    every iteration's trust-region step goes through promp_cg_solve, lowers the surrogate and keeps the mean KL inside the region"""
    import csv
    import importlib.util
    import os
    from promp_amd.utils import logger
    spec = importlib.util.spec_from_file_location('run_maml', os.path.join(scen.__file__.rsplit('/tests/', 1)[0], 'run_scripts', 'maml_run_point_mass.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = dict(mod.DEFAULT, n_itr=6, meta_batch_size=4, rollouts_per_meta_task=8, max_path_length=25, seed=5, exploration=exploration)
    logger.configure(dir=str(tmp_path), quiet=True)
    mod.main(cfg)
    rows = list(csv.DictReader(open(os.path.join(str(tmp_path), 'progress.csv'))))
    assert len(rows) == 6
    for r in rows:
        assert float(r['MeanKL']) <= 0.01 * 1.001, r['MeanKL']                      # trpo_maml.py:158: the trust region
        assert float(r['LossAfter']) <= float(r['LossBefore']) + 1e-9                # a rejected step restores the parameters
    assert any(float(r['LossAfter']) < float(r['LossBefore']) for r in rows)
    logger.configure(quiet=True)


def test_vpg_maml_on_device():
    scen.run_vpg_scenario(M=4, P=5, T=100, O=20, A=6, hidden=(64, 64), inner_type='log_likelihood')
    scen.run_vpg_scenario(M=3, P=3, T=40, O=5, A=3, hidden=(32, 32), inner_type='likelihood_ratio', exploration=True)


def test_trainer_snapshot_round_trip_on_device(tmp_path):
    scen.test_trainer_snapshot_round_trip(None, tmp_path)


def test_baseline_fit_predict_on_device():
    scen.run_baseline_fit_predict_scenario()


@pytest.mark.parametrize('name', ['default', 'ragged', 'raw', 'positive', 'retbase', 'retbase_raw'])
def test_dice_sample_processor_vs_reference_outputs(name):
    scen.run_dice_processor_scenario(name)


@pytest.mark.parametrize('name', ['k1_small', 'k1_ragged', 'k2_small', 'k1_hc', 'k1_long'])
def test_dice_maml_plugin(name):
    scen.run_dice_maml_scenario(name)


@pytest.mark.parametrize('name', ['k1_ragged', 'k2_small'])
def test_vpg_dice_maml_plugin(name):
    scen.run_vpg_dice_maml_scenario(name)
