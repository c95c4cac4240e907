#!/bin/bash
run() { echo "== $1"; shift; python bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['kernels']
print('  step %.3f ms | fwd_bwd %.1f us x%d  hvp %.1f us x%d  gram %.1f us'%(d['ms_per_step'],k['k_fwd_bwd']['avg_ms']*1e3,k['k_fwd_bwd']['launches_per_step'],k['k_hvp']['avg_ms']*1e3,k['k_hvp']['launches_per_step'],k['k_gram']['avg_ms']*1e3))"; }
run "full 40 tasks"
run "shard of 2 (20 tasks)" --shard-of 2
run "shard of 4 (10 tasks)" --shard-of 4
run "shard of 8 (5 tasks)" --shard-of 8
