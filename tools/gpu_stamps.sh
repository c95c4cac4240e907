#!/bin/bash
# developer tool (GPU box): per-region cycle stamps of k_pass / k_chain_hvp (tools/phase_timing.py) on a -DPROMP_DEV_STAMPS build
# (tools/build_variant.sh stamps -DPROMP_DEV_STAMPS); the product library is restored afterwards.
OUT=gpurun_out/stamps; mkdir -p $OUT; exec < /dev/null
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
cp tools/ablate/lib_stamps.so promp_amd/libpromp_hip.so
PROMP_STAMP_KERNELS=${1:-0,2} timeout 300 python tools/phase_timing.py > $OUT/stamps.txt 2>&1
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
cat $OUT/stamps.txt
