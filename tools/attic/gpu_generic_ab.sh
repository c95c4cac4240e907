#!/bin/bash
# generic shapes: parity tests of the layer-by-layer kernels, then tools/generic_timing.py (BF16 GEMMs; "fp32" as argument: also on the exact-FP32 ones)
mkdir -p gpurun_out/generic
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "generic or nonlinearit or humanoid or three_hidden or wide_observations" > gpurun_out/generic/pytest_generic.log 2>&1; echo "pytest generic rc=$?"
tail -4 gpurun_out/generic/pytest_generic.log
echo "== BF16 GEMMs"; python tools/generic_timing.py --steps 5 2>&1 | tee gpurun_out/generic/timing_bf16.txt
if [ "$1" = "fp32" ]; then echo "== exact-FP32 GEMMs (PROMP_GEN_FP32=1)"; PROMP_GEN_FP32=1 python tools/generic_timing.py --steps 5 2>&1 | tee gpurun_out/generic/timing_fp32.txt; fi
