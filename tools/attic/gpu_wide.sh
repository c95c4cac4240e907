#!/bin/bash
mkdir -p gpurun_out/wide
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "humanoid_width or sample_processing or fit_retry or full_size" > gpurun_out/wide/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" gpurun_out/wide/pytest.log | tail -5
timeout 300 python tools/stage_a_timing.py 40 111
timeout 300 python tools/stage_a_timing.py 40 20
timeout 600 python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --no-plugin-path > gpurun_out/wide/bench4.json 2> gpurun_out/wide/bench4.err; python -c "
import json; d=json.load(open('gpurun_out/wide/bench4.json')); print('config 4: %.3f ms/step' % d['ms_per_step']); print(d['roofline']['kernels']['k_gram'], d['roofline']['stage_a']['ms'])"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wide/trace -o w -- python tools/stage_a_timing.py 40 111 > /dev/null 2> gpurun_out/wide/trace.err; echo "trace rc=$?"
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/wide/trace/**/w_kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:8]:
    print('%-60s calls %6s avg %10.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
