#!/bin/bash
# developer tool: rocprofv3 kernel trace of a short bench run -> gpurun_out/trace/ (stats csv) and a printed summary
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
rm -rf $ROOT/gpurun_out/trace; mkdir -p $ROOT/gpurun_out/trace
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/trace -o trace -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline "$@" > $ROOT/gpurun_out/trace/bench.json 2> $ROOT/gpurun_out/trace/err.txt
cd $ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/trace/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    print('%-70s calls %5s  avg %9.1f us  total %6.2f %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
[ -n "$TIMELINE" ] && python tools/timeline.py gpurun_out/trace | tee gpurun_out/trace/timeline.txt
rm -f gpurun_out/trace/*/*kernel_trace.csv gpurun_out/trace/*kernel_trace.csv
