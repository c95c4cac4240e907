#!/bin/bash
# developer tool: one rank's share of the 40-task batch at 2 / 4 / 8 ranks, with k_pass and with k_pass_pair
mkdir -p gpurun_out
for rep in 1 2; do for n in 8 4 2; do for pp in 0 1; do
  PROMP_PASS_PAIR=$pp timeout 300 python bench.py --shard-of $n --steps 30 --warmup 3 --no-cpu-baseline --no-plugin-path --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d.get('roofline',{}).get('kernels',{})
print('shard-of $n pair=$pp: %d tasks, %.4f ms/step | ' % (d['config']['tasks_per_gpu'], d['ms_per_step']) + '  '.join('%s %.1f us' % (a, v['avg_ms']*1e3) for a, v in k.items()))"
done; done; done 2>&1 | tee gpurun_out/shard_pair.txt
