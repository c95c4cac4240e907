cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "hvp or primal_cache or meta or constraint" 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
cp promp_amd/libpromp_hip.so /tmp/prod.so; cp tools/ablate/lib_stamps.so promp_amd/libpromp_hip.so
PROMP_STAMP_KERNELS=1 timeout 200 python tools/phase_timing.py 2>&1 | tail -8
cp /tmp/prod.so promp_amd/libpromp_hip.so
