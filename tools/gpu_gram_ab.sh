#!/bin/bash
# developer tool: k_gram_tiled A/B -- config 4's bench line (k_gram per launch), Humanoid's step with double-buffered 8-row rounds
# against one 16-row tile (PROMP_GRAMT_SINGLE=1) and against k_gram_wide (PROMP_GRAM_UNTILED=1).  Output: gpurun_out/gram_ab/
exec < /dev/null
R=gpurun_out/gram_ab
rm -rf $R && mkdir -p $R
pick='import json,sys; d=json.loads(sys.stdin.read()); print("%.3f ms/step" % d["ms_per_step"], {k: round(v["avg_ms"]*1e3,1) for k,v in d["roofline"]["kernels"].items()})'
for v in "" "PROMP_GRAM_UNTILED=1"; do
  echo "config 4 $v: $(env $v timeout 120 python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --no-plugin-path 2>/dev/null | python -c "$pick")" | tee -a $R/ant.txt
done
for v in "" "PROMP_GRAMT_SINGLE=1"; do
  echo "== $v" | tee -a $R/humanoid.txt
  env $v timeout 200 python tools/generic_timing.py --case 2 --steps 4 2>&1 | head -6 | tee -a $R/humanoid.txt
done
