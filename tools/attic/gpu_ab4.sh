#!/bin/bash
# developer tool: A/B of the library variants under tools/ablate on BASELINE config 4 (Ant shapes, cooperative kernels)
REPS=${1:-2}
mkdir -p gpurun_out
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
for f in tools/ablate/lib_*.so; do
  cp $f promp_amd/libpromp_hip.so
  echo "== $f: tests"
  timeout 900 python -m pytest tests -m gpu -x -q -k "${PROMP_TEST_FILTER:-wide or ant or h128 or config}" 2>&1 | tail -2
done
for rep in $(seq $REPS); do
for f in tools/ablate/lib_*.so; do
  cp $f promp_amd/libpromp_hip.so
  echo "== $f"
  timeout 300 python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --no-plugin-path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  step %.4f ms | ' % d['ms_per_step'] + '  '.join('%s %.1f us' % (n, v['avg_ms']*1e3) for n, v in k.items()))"
done
done 2>&1 | tee gpurun_out/ab4.txt
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
