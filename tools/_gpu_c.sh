cd $GRAFT_REPO_ROOT
cp tools/ablate/lib_stamps.so promp_amd/libpromp_hip.so
PROMP_STAMP_KERNELS=2 timeout 200 python tools/phase_timing.py 2>&1 | tail -7
