#!/bin/bash
# developer tool: A/B bench of the library variants under tools/ablate, no tests (the product library is restored afterwards)
REPS=${1:-2}
mkdir -p gpurun_out
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
for rep in $(seq $REPS); do
for f in tools/ablate/lib_*.so; do
  cp $f promp_amd/libpromp_hip.so
  echo "== $f"
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-plugin-path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  step %.4f ms | ' % d['ms_per_step'] + '  '.join('%s %.1f us' % (n, v['avg_ms']*1e3) for n, v in k.items()))"
done
done 2>&1 | tee gpurun_out/ab.txt
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
