#!/bin/bash
# developer tool (GPU box): A/B of the library variants under tools/ablate on BASELINE config 4 (and config 3 with `3`)
OUT=gpurun_out/ab; mkdir -p $OUT; exec < /dev/null
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
for rep in $(seq ${1:-1}); do
for f in tools/ablate/lib_*.so; do
  cp $f promp_amd/libpromp_hip.so
  echo "== $f"
  timeout 300 python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --no-plugin-path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  config 4 step %.4f ms | ' % d['ms_per_step'] + str(d.get('fp16_split_events')) + ' ' + '  '.join('%s %.1f us' % (n, v['avg_ms']*1e3) for n, v in k.items()))"
done
done 2>&1 | tee $OUT/ab4.txt
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
