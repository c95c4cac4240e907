#!/bin/bash
# developer tool (GPU box): library variants under tools/ablate -- full-size parity, split accuracy at Ant's width, redone segments and
# step time on configs 3 and 4
OUT=gpurun_out/ab; mkdir -p $OUT; exec < /dev/null
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
for f in tools/ablate/lib_*.so; do
  cp $f promp_amd/libpromp_hip.so
  echo "== $f"
  timeout 600 python -m pytest tests -m gpu -q -s -k "full_config_meta_gradient or split_gemm_accuracy_guard" 2>&1 | grep -E "full-size parity|passed|failed"
  python tools/split_accuracy_gpu.py 2>&1 | grep "^(128\|^(64, 64) 20"
  for cfg in 3 4; do
  timeout 300 python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-plugin-path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  config $cfg step %.4f ms | ' % d['ms_per_step'] + str(d.get('fp16_split_events')) + ' ' + '  '.join('%s %.1f us' % (n, v['avg_ms']*1e3) for n, v in k.items()))"
  done
done 2>&1 | tee $OUT/vtarget.txt
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
