#!/bin/bash
# developer tool: per-kernel times of Stage A alone (stage_a_timing.py under rocprofv3 --kernel-trace --stats).  usage: gpu_stage_a_trace.sh [tasks obs_dim]
exec < /dev/null
R=gpurun_out/stage_a_trace
rm -rf $R && mkdir -p $R
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/t -o t -- python $ROOT/tools/stage_a_timing.py ${1:-40} ${2:-111} > $ROOT/$R/out.txt 2> $ROOT/$R/err.txt
echo "rc=$?"; cat $ROOT/$R/out.txt
f=$(find $ROOT/$R/t -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-4,6,7 "$f" | head -12 | tee $ROOT/$R/kernels.txt
find $ROOT/$R -name "*kernel_trace.csv" -delete
