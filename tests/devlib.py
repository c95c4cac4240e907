"""Which C-ABI library a test binds (test infrastructure).

* ``-m gpu`` tests bind the product library promp_amd/libpromp_hip.so on a real MI355X.
* ``-m "not gpu"`` tests may bind tests/emu/libpromp_emu.so: the SAME kernel and host sources compiled by
  g++ against the SIMT interpreter tests/emu/hip_emu.h, to verify kernel indexing and host sequencing in
  the GPU-less build container.  It is slow (one OS thread per lane) so only tiny shapes are used.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from promp_amd import _lib  # noqa: E402

_emu = None


def emu_library():
    global _emu
    if _emu is None:
        sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
        import build_emu
        _emu = _lib.Library(build_emu.build())
    return _emu


def gpu_library():
    return _lib.Library()   # raises if libpromp_hip.so is missing: no fallback
