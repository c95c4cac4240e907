#!/bin/bash
# developer tool: A/B bench of the library variants under tools/ablate (the product library is restored afterwards)
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for f in tools/ablate/lib_*.so; do
  cp $f promp_amd/libpromp_hip.so
  echo "== $f"
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  step %.4f ms | ' % d['ms_per_step'] + '  '.join('%s %.1f us' % (n, v['avg_ms']*1e3) for n, v in k.items()))"
  python bench.py --shard-of 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  shard8 ms %.4f' % d['ms_per_step'])"
done
done
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
