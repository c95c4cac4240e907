#!/bin/bash
run() { echo "== $1"; shift; env "$@" python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  %.1fM env-steps/s  %.3f ms/step | fwd_bwd %.1f us (%.1f TF)  hvp %.1f us (%.1f TF)  gram %.1f us'%(d['value']/1e6,d['ms_per_step'],k['k_fwd_bwd']['avg_ms']*1e3,k['k_fwd_bwd']['tflops'],k['k_hvp']['avg_ms']*1e3,k['k_hvp']['tflops'],k['k_gram']['avg_ms']*1e3))"; }
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run "8 waves (default)" A=1
run "4 waves" PROMP_DEV_FWD_WAVES=4
run "8 waves, target0 128" PROMP_DEV_TARGET0=128
