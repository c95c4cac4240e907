#!/bin/bash
# developer tool (GPU box): per-launch timeline of one config-4 step (both streams)
OUT=gpurun_out/tl4; rm -rf $OUT; mkdir -p $OUT; exec < /dev/null
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/trace4 -o trace4 -- python $ROOT/bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-plugin-path > /dev/null 2> $ROOT/$OUT/trace4.err
cd $ROOT
TIMELINE_STEP=${1:-2} python tools/timeline.py $OUT/trace4 > $OUT/timeline_config4.txt 2>&1
rm -rf $OUT/trace4
head -${2:-16} $OUT/timeline_config4.txt; tail -1 $OUT/timeline_config4.txt
