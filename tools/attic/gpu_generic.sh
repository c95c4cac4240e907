#!/bin/bash
# the generic-shape path on the GPU: its parity tests, then the whole GPU suite, then timings
mkdir -p gpurun_out/generic
timeout 900 python -m pytest tests -m gpu -x -q -k "generic or unsupported or wide_observations or get_actions or abi" > gpurun_out/generic/pytest_generic.log 2>&1; echo "pytest generic rc=$?"
tail -15 gpurun_out/generic/pytest_generic.log
timeout 600 python tools/generic_timing.py > gpurun_out/generic/timing.txt 2> gpurun_out/generic/timing.err; echo "timing rc=$?"
cat gpurun_out/generic/timing.txt; tail -5 gpurun_out/generic/timing.err
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/generic/pytest_gpu.log 2>&1; echo "pytest all rc=$?"
tail -6 gpurun_out/generic/pytest_gpu.log
