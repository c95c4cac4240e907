#!/bin/bash
mkdir -p gpurun_out/wide
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "humanoid_width or sample_processing or generic or fit_retry" > gpurun_out/wide/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" gpurun_out/wide/pytest.log | tail -5
timeout 300 python tools/stage_a_timing.py 40 111
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wide/trace -o w -- python tools/generic_timing.py --steps 3 > gpurun_out/wide/timing.txt 2> gpurun_out/wide/trace.err; echo "trace rc=$?"
grep -E "Humanoid|gram" gpurun_out/wide/timing.txt
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/wide/trace/**/w_kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0]))):
    if any(k in r['Name'] for k in ('k_gram', 'k_fit', 'k_gae', 'k_returns')):
        print('%-60s calls %6s avg %10.1f us  total %8.2f ms' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
