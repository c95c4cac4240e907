#!/bin/bash
# developer tool: parity tests (filtered) + A/B bench + shard-of-8 of every library variant under tools/ablate; runs in a scratch copy of
# the product library path so that nothing is left behind
REPS=${1:-2}
mkdir -p gpurun_out
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
for f in tools/ablate/lib_*.so; do
  cp $f promp_amd/libpromp_hip.so
  echo "== $f: tests"
  timeout 900 python -m pytest tests -m gpu -x -q -k "${PROMP_TEST_FILTER:-cache or hvp or meta or guard}" 2>&1 | tail -2
done
for rep in $(seq $REPS); do
for f in tools/ablate/lib_*.so; do
  cp $f promp_amd/libpromp_hip.so
  echo "== $f"
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-plugin-path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  step %.4f ms | ' % d['ms_per_step'] + '  '.join('%s %.1f us' % (n, v['avg_ms']*1e3) for n, v in k.items()))"
  timeout 300 python bench.py --shard-of 8 --steps 30 --warmup 3 --no-cpu-baseline --no-plugin-path --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d.get('roofline',{}).get('kernels',{})
print('  shard-of 8: %.4f ms/step | ' % d['ms_per_step'] + '  '.join('%s %.1f us' % (a, v['avg_ms']*1e3) for a, v in k.items()))"
done
done 2>&1 | tee gpurun_out/ab.txt
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
