#!/bin/bash
# developer tool: A/B bench of the library variants under tools/ablate (the product library is restored afterwards)
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
for f in tools/ablate/lib_*.so; do
  cp $f promp_amd/libpromp_hip.so
  echo "== $f"
  python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  step %.3f ms | ' % d['ms_per_step'] + '  '.join('%s %.1f us' % (n, v['avg_ms']*1e3) for n, v in k.items()))"
done
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
