#!/bin/bash
mkdir -p gpurun_out/generic
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "generic or unsupported or wide_observations or get_actions or abi" > gpurun_out/generic/pytest_generic.log 2>&1; echo "pytest generic rc=$?"
tail -8 gpurun_out/generic/pytest_generic.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/generic/trace -o gen -- python tools/generic_timing.py --steps 3 > gpurun_out/generic/timing_prof.txt 2> gpurun_out/generic/trace.err; echo "trace rc=$?"
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/generic/trace/**/gen_kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:25]:
    print('%-70s calls %6s avg %10.1f us  total %8.2f ms %5s%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
