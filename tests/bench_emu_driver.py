"""Test infrastructure: bench.py with the kernel-emulation library bound (tests/emu/libpromp_emu.so) instead of the product's.
tests/test_bench_ranks.py starts it with --gpus 2: bench.py then launches its two ranks itself (this file again, through
promp_amd.launch), which exchange through the emulator's shared-memory RCCL shim -- the N-rank code path of bench.py (setup,
run_timed, the rccl block, weak_batch) executes before hardware ever sees it.  Never a benchmark."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from promp_amd import _lib  # noqa: E402
from tests import devlib  # noqa: E402

_lib.set_library_for_testing(devlib.emu_library())
import bench  # noqa: E402

if __name__ == '__main__':
    bench.main()
