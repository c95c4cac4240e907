#!/bin/bash
# developer tool: Stage A alone (config 3 and Ant's width) + the sample-processing parity tests + configs 3 / 4 step times.
exec < /dev/null
R=gpurun_out/stage_a_ab
rm -rf $R && mkdir -p $R
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plugin_api.py -m gpu -q -x -k "sample_processing or humanoid_width or fit_ or baseline or predict or plugin" 2>&1 | grep -E "passed|failed" | tee $R/pytest.txt
for rep in 1 2; do
  timeout 100 python tools/stage_a_timing.py 2>&1 | tee -a $R/stage_a.txt
  timeout 100 python tools/stage_a_timing.py 40 111 2>&1 | tee -a $R/stage_a.txt
done
for cfg in 3 4; do
  echo "config $cfg: $(timeout 120 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-plugin-path --repeats 1 2>/dev/null | python -c 'import json,sys; print("%.4f ms/step" % json.loads(sys.stdin.read())["ms_per_step"])')" | tee -a $R/steps.txt
done
