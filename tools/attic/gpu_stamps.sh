#!/bin/bash
# developer tool: cycle stamps of k_pass / k_chain_hvp with the -DPROMP_DEV_STAMPS library variant
mkdir -p gpurun_out
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
cp tools/ablate/lib_stamps.so promp_amd/libpromp_hip.so
python tools/phase_timing.py > gpurun_out/phase_timing.txt 2>&1
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
cat gpurun_out/phase_timing.txt
