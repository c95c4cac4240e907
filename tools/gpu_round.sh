#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprofv3 kernel trace.  Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
timeout 600 python bench.py --steps 30 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err); echo "rocprof rc=$?"
find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -20 "$f"
