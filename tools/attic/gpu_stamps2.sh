#!/bin/bash
# developer tool: cycle stamps with the -DPROMP_DEV_STAMPS library variant in tools/stamps/
mkdir -p gpurun_out
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
cp tools/stamps/lib_stamps.so promp_amd/libpromp_hip.so
PROMP_STAMP_KERNELS=${PROMP_STAMP_KERNELS:-2} python tools/phase_timing.py > gpurun_out/phase_timing.txt 2>&1
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
cat gpurun_out/phase_timing.txt
