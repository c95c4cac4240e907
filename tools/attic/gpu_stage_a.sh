#!/bin/bash
# developer tool: Stage A (process_samples) standalone -- per-kernel durations (rocprofv3) and cycle stamps (-DPROMP_DEV_STAMPS variant)
mkdir -p gpurun_out/stage_a
export TMPDIR=/tmp
python tools/stage_a_timing.py > gpurun_out/stage_a/timing.txt 2>&1; cat gpurun_out/stage_a/timing.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stage_a/trace -o sa -- python tools/stage_a_timing.py > /dev/null 2> gpurun_out/stage_a/trace.err
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/stage_a/trace/**/sa_kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:8]:
    print('%-50s calls %6s avg %8.1f us' % (r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3))
PY
if [ -f tools/ablate/lib_stamps.so ]; then
  cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
  cp tools/ablate/lib_stamps.so promp_amd/libpromp_hip.so
  python tools/stage_a_timing.py 2>&1 | grep cycles | sort | awk "{k=\$1\" \"\$2; c[k]++; if (c[k]<=4) print}" > gpurun_out/stage_a/stamps.txt
  cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
  cat gpurun_out/stage_a/stamps.txt
fi
