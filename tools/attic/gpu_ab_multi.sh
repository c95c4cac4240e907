#!/bin/bash
# developer tool: several library builds (tools/ablate/lib_<name>.so) against each other on one box, configs 3 and 4, alternating
# usage: bash tools/attic/gpu_ab_multi.sh name1 name2 ...
mkdir -p gpurun_out/ab
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
for rep in 1 2; do for v in "$@"; do
  cp tools/ablate/lib_$v.so promp_amd/libpromp_hip.so
  for cfg in 3 4; do
  python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-plugin-path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-12s rep $rep config $cfg: %.3f ms/step ' % ('$v', d['ms_per_step']), {k:round(x['avg_ms']*1e3,1) for k,x in d['roofline']['kernels'].items() if 'gram' not in k})"
  done
done; done | tee gpurun_out/ab/ab_multi.txt
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
