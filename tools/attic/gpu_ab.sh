#!/bin/bash
# developer tool: GPU parity tests on the product library, then an A/B bench of the library variants under tools/ablate
# (the product library is restored afterwards).  usage: tools/gpu_ab.sh [reps]
REPS=${1:-1}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
for rep in $(seq $REPS); do
for f in tools/ablate/lib_*.so; do
  cp $f promp_amd/libpromp_hip.so
  echo "== $f"
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  step %.4f ms | ' % d['ms_per_step'] + '  '.join('%s %.1f us' % (n, v['avg_ms']*1e3) for n, v in k.items()))"
  timeout 300 python bench.py --shard-of 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  shard8 ms %.4f' % d['ms_per_step'])"
done
done 2>&1 | tee gpurun_out/ab.txt
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
