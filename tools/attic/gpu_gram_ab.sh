#!/bin/bash
# developer tool: k_gram_tiled A/B -- parity of the wide baseline fits, config 4's bench line (k_gram per launch) against k_gram_wide
# (PROMP_GRAM_UNTILED=1).  Output: gpurun_out/gram_ab/
exec < /dev/null
R=gpurun_out/gram_ab
rm -rf $R && mkdir -p $R
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sample_processing or humanoid_width or fit_" 2>&1 | grep -E "passed|failed" | tee $R/pytest.txt
pick='import json,sys; d=json.loads(sys.stdin.read()); print("%.3f ms/step" % d["ms_per_step"], {k: round(v["avg_ms"]*1e3,1) for k,v in d["roofline"]["kernels"].items()})'
for rep in 1 2; do for v in "PROMP_X=0" "PROMP_GRAM_UNTILED=1"; do
  echo "config 4 $v: $(env $v timeout 120 python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --no-plugin-path 2>/dev/null | python -c "$pick")" | tee -a $R/ant.txt
done; done
