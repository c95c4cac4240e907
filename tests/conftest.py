import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _inside_a_worker(config):
    """True in a pytest-xdist worker (and in anything a parallel run of this suite started): the hook below must never fan out there --
    a worker that starts workers of its own is a fork bomb."""
    return bool(os.environ.get('PYTEST_XDIST_WORKER')) or hasattr(config, 'workerinput') or bool(os.environ.get('PROMP_TESTS_PARENT_PID'))


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """CPU suite (no GPU in the box): about 70 of its tests run the kernel sources on the thread-per-lane emulator and take most of
    the suite's half hour one after the other -- share the tests out over a few worker processes (pytest-xdist, when installed and
    the caller gave no -n).  On a GPU box the tests stay in one process: one context at a time on the device.
    PROMP_TESTS_SERIAL=1 keeps one process here too."""
    if have_gpu() or _inside_a_worker(config) or os.environ.get('PROMP_TESTS_SERIAL'):
        return None
    if not config.pluginmanager.hasplugin('xdist') or getattr(config.option, 'numprocesses', None):
        return None
    if getattr(config.option, 'collectonly', False) or getattr(config.option, 'usepdb', False):
        return None
    os.environ['PROMP_TESTS_PARENT_PID'] = str(os.getpid())          # inherited by the workers: second guard beside xdist's own variable
    config.option.numprocesses = max(2, min(4, (os.cpu_count() or 2) // 2))
    config.option.dist = 'load'
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def have_gpu():
    return os.path.exists('/dev/kfd')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
