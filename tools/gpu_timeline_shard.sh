#!/bin/bash
# developer tool (GPU box): per-launch timeline of one step of a 1-of-N shard of config 3 (default N = 8: 5 tasks on this GPU)
N=${1:-8}
OUT=gpurun_out/tls; rm -rf $OUT; mkdir -p $OUT; exec < /dev/null
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/trace -o t -- python $ROOT/bench.py --shard-of $N --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-plugin-path --repeats 1 > /dev/null 2> $ROOT/$OUT/trace.err
cd $ROOT
TIMELINE_STEP=${2:-6} python tools/timeline.py $OUT/trace > $OUT/timeline.txt 2>&1
rm -rf $OUT/trace
cat $OUT/timeline.txt
