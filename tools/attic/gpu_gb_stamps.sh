#!/bin/bash
# developer tool: cycles per phase of k_gb_linear with the -DPROMP_DEV_STAMPS library variant (tools/build_variant.sh stamps -DPROMP_DEV_STAMPS)
mkdir -p gpurun_out
cp promp_amd/libpromp_hip.so /tmp/lib_keep.so
cp tools/ablate/lib_stamps.so promp_amd/libpromp_hip.so
python tools/generic_timing.py --steps 1 --case ${1:-1} > gpurun_out/gb_stamps_raw.txt 2>&1
cp /tmp/lib_keep.so promp_amd/libpromp_hip.so
grep "k_gb_linear" gpurun_out/gb_stamps_raw.txt | sort | uniq -c | sort -rn | head -40
