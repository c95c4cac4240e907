#!/bin/bash
# developer tool: cycle stamps of the BF16-pipe cooperative kernels with the -DPROMP_DEV_STAMPS library variant
mkdir -p gpurun_out
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
cp tools/ablate/lib_stamps.so promp_amd/libpromp_hip.so
python tools/wb_stamps.py ${1:-0} > gpurun_out/wb_stamps.txt 2>&1
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
cat gpurun_out/wb_stamps.txt
