#!/bin/bash
# developer tool: A/B of two library builds on the same box: tools/ablate/lib_<A>.so against lib_<B>.so, config 4, alternating
# usage: bash tools/attic/gpu_ab_wb.sh A B [config]
A=$1; B=$2; CFG=${3:-4}
mkdir -p gpurun_out/ab
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do for v in $A $B; do
  cp tools/ablate/lib_$v.so promp_amd/libpromp_hip.so
  python bench.py --config $CFG --steps 10 --warmup 2 --no-cpu-baseline --no-plugin-path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v rep $rep: %.3f ms/step ' % d['ms_per_step'], {k:round(x['avg_ms']*1e3,1) for k,x in d['roofline']['kernels'].items() if 'gram' not in k})"
done; done | tee gpurun_out/ab/ab_${A}_${B}.txt
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
