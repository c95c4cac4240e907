#!/bin/bash
# developer tool: BF16-pipe cooperative kernels: stamps of both kernels, config-4 bench, parity on 128-wide shapes
mkdir -p gpurun_out/wb1
bash tools/gpu_wb_stamps.sh 0,1
timeout 600 python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --no-plugin-path > gpurun_out/wb1/b.json 2> gpurun_out/wb1/b.err; echo "bench rc=$?"; tail -2 gpurun_out/wb1/b.err
python -c "
import json
d=json.load(open('gpurun_out/wb1/b.json'))
print('config4 ms/step %.3f  value %.2f M' % (d['ms_per_step'], d['value']/1e6)); print({k:round(v['avg_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "128 or config4 or split or any_hidden or shape or trpo or constraint" > gpurun_out/wb1/pytest.log 2>&1; tail -4 gpurun_out/wb1/pytest.log
