#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command -> gpurun_out/trace/
export TMPDIR=/tmp
rm -rf gpurun_out/trace && mkdir -p gpurun_out/trace
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/trace/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/trace/err.txt
echo rc=$?
ls $GRAFT_REPO_ROOT/gpurun_out/trace
cat $GRAFT_REPO_ROOT/gpurun_out/trace/*kernel_stats.csv | head -30
