#!/bin/bash
# developer tool: split accuracy at 128-wide, the guard test, config-4 bench and stamps
mkdir -p gpurun_out/wb4
python tools/split_accuracy_gpu.py > gpurun_out/wb4/split_accuracy.txt 2>&1; tail -3 gpurun_out/wb4/split_accuracy.txt
PROMP_WIDE_FP32=1 python tools/split_accuracy_gpu.py 2>&1 | tail -2 | sed 's/^/fp32 kernels: /' >> gpurun_out/wb4/split_accuracy.txt; tail -2 gpurun_out/wb4/split_accuracy.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "split" 2>&1 | tail -2
bash tools/gpu_wb_stamps.sh 0,1 | grep "round total\|extra"
timeout 600 python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --no-plugin-path > gpurun_out/wb4/b4.json 2> gpurun_out/wb4/b4.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('gpurun_out/wb4/b4.json'))
print('config4 ms/step %.3f  value %.2f M' % (d['ms_per_step'], d['value']/1e6)); print({k:round(v['avg_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})"
