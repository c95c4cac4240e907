#!/bin/bash
# developer tool (GPU box): split-accuracy figures and the parity tests of the product library, then an A/B bench of the library
# variants under tools/ablate (tools/build_variant.sh); the product library is restored afterwards.  usage: tools/gpu_ab.sh [reps] [pytest -k]
REPS=${1:-1}
KSEL=${2:-}
OUT=gpurun_out/ab
mkdir -p $OUT
exec < /dev/null
timeout 300 python tools/split_accuracy_gpu.py > $OUT/split_accuracy.txt 2>&1; tail -12 $OUT/split_accuracy.txt
if [ -n "$KSEL" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$KSEL" > $OUT/pytest_gpu.log 2>&1; else timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; fi
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
for rep in $(seq $REPS); do
for f in tools/ablate/lib_*.so; do
  cp $f promp_amd/libpromp_hip.so
  echo "== $f"
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-plugin-path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  step %.4f ms | ' % d['ms_per_step'] + str(d.get('fp16_split_events')) + ' ' + '  '.join('%s %.1f us' % (n, v['avg_ms']*1e3) for n, v in k.items()))"
done
done 2>&1 | tee $OUT/ab.txt
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
