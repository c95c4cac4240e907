#!/bin/bash
# developer tool: shader clock / power while the bench loop runs (is the step power- or clock-limited?)
mkdir -p gpurun_out
python bench.py --steps 4000 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
BP=$!
sleep 4
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr '\n' ' '
  echo
  sleep 0.5
done > gpurun_out/clock_probe.txt
wait $BP
cat gpurun_out/clock_probe.txt
python -c "
import json; d=json.load(open('gpurun_out/clock_bench.json')); print('ms/step', d['ms_per_step'], 'value', d['value'])"
