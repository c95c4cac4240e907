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
print('  step %.3f ms | fwd_bwd %.1f us  hvp %.1f us  fwd %.1f us'%(d['ms_per_step'],k['k_fwd_bwd']['avg_ms']*1e3,k['k_hvp']['avg_ms']*1e3,k['k_fwd_bwd<fwd-only>']['avg_ms']*1e3))"
done
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
