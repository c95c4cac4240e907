#!/bin/bash
# developer tool: per-launch timeline of one config-4 step (both streams) from a rocprofv3 kernel trace.  Output: gpurun_out/tl4/
exec < /dev/null
R=gpurun_out/tl4
rm -rf $R && mkdir -p $R
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
env $1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$R/trace -o t -- python $ROOT/bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-plugin-path --repeats 1 > $ROOT/$R/bench.json 2> $ROOT/$R/err.txt
echo "rc=$?"
cd $ROOT
TIMELINE_STEP=${2:-4} python tools/timeline.py $R/trace > $R/timeline.txt 2>&1
head -30 $R/timeline.txt; tail -3 $R/timeline.txt
find $R -name "*kernel_trace.csv" -delete
