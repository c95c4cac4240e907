#!/bin/bash
# per-kernel profile of the layer-by-layer kernels (generic shapes): rocprofv3 kernel stats over tools/generic_timing.py
mkdir -p gpurun_out/generic
export TMPDIR=/tmp
python tools/generic_timing.py --steps 5 > gpurun_out/generic/timing.txt 2>&1; echo "timing rc=$?"
cat gpurun_out/generic/timing.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/generic/trace -o gen -- python tools/generic_timing.py --steps 3 > gpurun_out/generic/timing_prof.txt 2> gpurun_out/generic/trace.err; echo "trace rc=$?"
python tools/gen_trace_cases.py gpurun_out/generic/trace/gen_kernel_trace.csv
