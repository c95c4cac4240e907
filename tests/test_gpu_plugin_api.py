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


def test_policy_algo_api_two_inner_steps():
    scen.run_algo_scenario(M=3, P=4, T=50, O=5, A=3, hidden=(32, 32), K=2, epochs=3)


def test_trainer_end_to_end_point_env():
    scen.run_trainer_scenario(n_itr=3)


def test_get_actions_on_device():
    scen.run_get_actions_scenario(M=8, B=20, O=20, A=6, hidden=(64, 64))
