"""CPU checks of the drop-in boundary: the header, the binding table and the built libraries agree."""
import os
import re

import pytest

from promp_amd import _lib
from tests import devlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'promp_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(promp_[a-z0-9_]+)\s*\(', txt)))


def test_binding_table_matches_header():
    assert header_symbols() == sorted(_lib.SIGNATURES.keys())


def test_emulated_library_exports_every_symbol():
    lib = devlib.emu_library()
    for s in header_symbols():
        assert hasattr(lib.cdll, s)
    assert lib.cdll.promp_abi_version() == 3


@pytest.mark.skipif(not os.path.exists(_lib.DEFAULT_LIBRARY), reason='libpromp_hip.so not built (run __graft_entry__.build())')
def test_product_library_loads_and_exports_every_symbol():
    lib = _lib.Library()           # hipcc-built gfx950 library; loads without a GPU
    for s in header_symbols():
        assert hasattr(lib.cdll, s)
    d = _lib.Dims(40, 40, 20, 6, 64, 64, 1, 1000, 10)
    assert lib.cdll.promp_param_count(d) == 5900          # Theta at config 3
    assert lib.cdll.promp_feature_dim(d, 1) == 44          # D = 2*O+4


@pytest.mark.skipif(not os.path.exists(_lib.DEFAULT_LIBRARY) or os.path.exists('/dev/kfd'),
                    reason='needs the built library and NO gpu')
def test_no_cpu_fallback_without_a_gpu():
    with pytest.raises(_lib.PrompError, match='no HIP device|no CPU fallback'):
        _lib.Context(2, 4, 2, (32, 32), 1, max_rows=10, max_paths=2)


def test_argument_errors_are_reported():
    lib = devlib.emu_library()
    with pytest.raises(_lib.PrompError, match='obs_dim'):
        _lib.Context(2, 2000, 8, (64, 64), 1, max_rows=10, max_paths=2, lib=lib)
    with pytest.raises(_lib.PrompError, match='hidden'):
        _lib.Context(2, 4, 2, (512, 256), 1, max_rows=10, max_paths=2, lib=lib)
    with pytest.raises(_lib.PrompError, match='hidden'):
        _lib.Context(2, 4, 2, (64, 64, 64, 64, 64), 1, max_rows=10, max_paths=2, lib=lib)
    _lib.Context(2, 200, 17, (64, 129, 256), 1, max_rows=10, max_paths=2, lib=lib).close()   # layer-by-layer kernels (ABI 3)
    _lib.Context(2, 4, 2, (64, 128), 1, max_rows=10, max_paths=2, lib=lib).close()     # runs zero-padded on (128, 128)
    _lib.Context(2, 4, 2, (64, 32), 1, max_rows=10, max_paths=2, lib=lib).close()      # every combination of {32, 64}
    ctx = _lib.Context(2, 4, 2, (32, 32), 1, max_rows=10, max_paths=2, lib=lib)
    with pytest.raises(_lib.PrompError, match='no data'):
        ctx.process_samples(0)
    with pytest.raises(_lib.PrompError, match='discount'):
        import numpy as np
        ctx.upload_step(0, [0, 1, 2], [0, 3, 6], np.zeros((6, 4)), np.zeros(6))
        ctx.process_samples(0, discount=1.5)
