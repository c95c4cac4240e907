#!/bin/bash
run() { echo "== $1"; shift; env "$@" python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  step %.3f ms | fwd_bwd %.1f us x%d  hvp %.1f us x%d  gram %.1f us'%(d['ms_per_step'],k['k_fwd_bwd']['avg_ms']*1e3,k['k_fwd_bwd']['launches_per_step'],k['k_hvp']['avg_ms']*1e3,k['k_hvp']['launches_per_step'],k['k_gram']['avg_ms']*1e3))"; }
run "8 waves" A=1
run "12 waves" PROMP_DEV_FWD_WAVES=12
