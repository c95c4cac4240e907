#!/bin/bash
# developer tool: GPU tests (fail fast), then the bench with the primal cache on / off, full batch and the 5-task shard
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
for rep in 1 2; do
for flag in "" "--no-primal-cache"; do
  echo "== cache flag: '$flag'"
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline $flag 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  step %.4f ms  %.1f M/s | ' % (d['ms_per_step'], d['value']/1e6) + '  '.join('%s %.1f us' % (n, v['avg_ms']*1e3) for n, v in k.items()))"
  python bench.py --shard-of 8 --no-cpu-baseline --no-roofline $flag 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  shard8 ms %.4f' % d['ms_per_step'])"
done
done
