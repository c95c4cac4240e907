#!/bin/bash
mkdir -p gpurun_out
timeout 120 python tools/pair_debug.py 2>&1 | tail -7
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
cp tools/ablate/lib_stamps.so promp_amd/libpromp_hip.so
PROMP_STAMP_KERNELS=0 timeout 120 python tools/phase_timing.py > gpurun_out/phase_timing.txt 2>&1
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
cat gpurun_out/phase_timing.txt
for f in tools/ablate/lib_*.so; do
  case $f in *stamps*) continue;; esac
  cp $f promp_amd/libpromp_hip.so
  echo "== $f"
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  step %.4f ms | ' % d['ms_per_step'] + '  '.join('%s %.1f us' % (n, v['avg_ms']*1e3) for n, v in k.items()))"
done
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
