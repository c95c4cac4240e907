#!/bin/bash
# Humanoid-width baseline fit: parity tests, then the Humanoid case of tools/generic_timing.py with one launch per phase and with k_fit_wide alone
mkdir -p gpurun_out/generic
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "humanoid_width or fit_retries or one_launch_per_phase or sample_processing" > gpurun_out/generic/pytest_fit.log 2>&1; echo "pytest fit rc=$?"
tail -4 gpurun_out/generic/pytest_fit.log
echo "== one launch per phase"; python tools/generic_timing.py --steps 5 --case 2 2>&1 | tee gpurun_out/generic/timing_fit_phases.txt
echo "== k_fit_wide alone (PROMP_FIT_ONE_LAUNCH=1)"; PROMP_FIT_ONE_LAUNCH=1 python tools/generic_timing.py --steps 5 --case 2 2>&1 | tee gpurun_out/generic/timing_fit_one.txt
python tools/stage_a_timing.py 2>&1 | tail -6
