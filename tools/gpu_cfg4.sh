#!/bin/bash
# developer tool: config-4 (Ant shapes) checks: wide sample-processing parity tests, Stage A alone, the bench line, kernel stats
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "sample_processing or config4" 2>&1 | tail -3
python tools/stage_a_timing.py 40 111
python tools/stage_a_timing.py 40 20
timeout 600 python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --no-plugin-path > gpurun_out/bench_config4.json 2> gpurun_out/bench_config4.err; echo "rc=$?"; tail -2 gpurun_out/bench_config4.err
python -c "
import json
d=json.load(open('gpurun_out/bench_config4.json'))
print('config4 ms/step %.3f  value %.2f M' % (d['ms_per_step'], d['value']/1e6)); print({k:round(v['avg_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()}, d['roofline'].get('stage_a',{}).get('ms'))"
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace4 -o trace4 -- python $R/bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-plugin-path > /dev/null 2> $R/gpurun_out/trace4.err
cd $R; python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/trace4/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]: print('%-60s calls %5s avg %10.1f us  %5.1f %%' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
rm -f gpurun_out/trace4/*/*kernel_trace.csv
