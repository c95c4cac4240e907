#!/bin/bash
run() { echo "== $1"; shift; env "$@" python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  %.1fM env-steps/s  %.3f ms/step | fwd_bwd %.1f us (%.1f TF)  hvp %.1f us (%.1f TF)  gram %.1f us'%(d['value']/1e6,d['ms_per_step'],k['k_fwd_bwd']['avg_ms']*1e3,k['k_fwd_bwd']['tflops'],k['k_hvp']['avg_ms']*1e3,k['k_hvp']['tflops'],k['k_gram']['avg_ms']*1e3))"; }
run "default (2 WG/CU, 512 target)" A=1
run "1 WG/CU, target 256" PROMP_DEV_FWD_LDS_PAD=12000 PROMP_DEV_FWD_TARGET=256
run "1 WG/CU, target 512" PROMP_DEV_FWD_LDS_PAD=12000 PROMP_DEV_FWD_TARGET=512
run "2 WG/CU, target 256" PROMP_DEV_FWD_TARGET=256
run "2 WG/CU, target 1024" PROMP_DEV_FWD_TARGET=1024
