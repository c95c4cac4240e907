#!/bin/bash
# developer tool: full GPU suite + config 3 / 4 bench lines + stamps of the BF16-pipe cooperative kernels
mkdir -p gpurun_out/wb3
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/wb3/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/wb3/pytest_gpu.log
bash tools/gpu_wb_stamps.sh 0,1 | head -34
for cfg in 4 3; do
timeout 600 python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-plugin-path > gpurun_out/wb3/b$cfg.json 2> gpurun_out/wb3/b$cfg.err; echo "bench rc=$?"; tail -2 gpurun_out/wb3/b$cfg.err
python -c "
import json
d=json.load(open('gpurun_out/wb3/b$cfg.json'))
print('config$cfg ms/step %.3f  value %.2f M' % (d['ms_per_step'], d['value']/1e6)); print({k:round(v['avg_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})"
done
