#!/bin/bash
# developer tool: priority of the second stream (sample processing of steps >= 1) -- high (default), low, none; configs 3 and 4
exec < /dev/null
mkdir -p gpurun_out/prio
for rep in 1 2; do for v in "PROMP_SIDE_PRIO=hi" "PROMP_SIDE_PRIO=lo" "PROMP_SIDE_PRIO=none"; do for cfg in 3 4; do
  echo "$v rep $rep config $cfg: $(env $v timeout 120 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-plugin-path --repeats 1 2>/dev/null | python -c 'import json,sys; print("%.4f ms/step" % json.loads(sys.stdin.read())["ms_per_step"])')" | tee -a gpurun_out/prio/ab.txt
done; done; done
