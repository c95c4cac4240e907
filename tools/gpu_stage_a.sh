#!/bin/bash
# developer tool (GPU box): Stage A alone at Ant's width (40 tasks, obs_dim 111) -- wall time of one chain, per-kernel averages of the
# chain run back to back on ONE stream (rocprofv3 kernel trace), and k_fit_wide's phase stamps on a -DPROMP_DEV_STAMPS build
# (tools/build_variant.sh stamps -DPROMP_DEV_STAMPS) when tools/ablate/lib_stamps.so is there.
OUT=gpurun_out/stage_a; rm -rf $OUT; mkdir -p $OUT; exec < /dev/null
export TMPDIR=/tmp
ROOT=$PWD
M=${1:-40}; O=${2:-111}
timeout 300 python tools/stage_a_timing.py $M $O > $OUT/wall.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -o t -- python $ROOT/tools/stage_a_timing.py $M $O > /dev/null 2> $ROOT/$OUT/trace.err
cd $ROOT
python - <<PY > $OUT/kernels.txt
import csv, glob
f = glob.glob('$OUT/trace/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    print('%-70s calls %5s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
rm -rf $OUT/trace
if [ -f tools/ablate/lib_stamps.so ]; then
  cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
  cp tools/ablate/lib_stamps.so promp_amd/libpromp_hip.so
  timeout 300 python tools/stage_a_timing.py $M $O 2>&1 | grep -E "cycles" | sort | uniq -c | sort -rn | head -12 > $OUT/stamps.txt
  cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
fi
cat $OUT/wall.txt $OUT/kernels.txt $OUT/stamps.txt
