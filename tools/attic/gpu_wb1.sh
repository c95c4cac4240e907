#!/bin/bash
# developer tool: first GPU contact of the BF16-pipe cooperative kernels: parity on 128-wide shapes, config-4 bench A/B
mkdir -p gpurun_out/wb1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "128 or config4 or split or any_hidden or shape" 2>&1 | tail -5
for mode in 0 1; do
  PROMP_WIDE_FP32=$mode timeout 600 python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --no-plugin-path > gpurun_out/wb1/bench_config4_fp32_$mode.json 2> gpurun_out/wb1/bench_$mode.err; echo "rc=$?"; tail -2 gpurun_out/wb1/bench_$mode.err
  python -c "
import json
d=json.load(open('gpurun_out/wb1/bench_config4_fp32_$mode.json'))
print('PROMP_WIDE_FP32=$mode config4 ms/step %.3f  value %.2f M' % (d['ms_per_step'], d['value']/1e6)); print({k:round(v['avg_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()}, d['roofline'].get('stage_a',{}).get('ms'))"
done
