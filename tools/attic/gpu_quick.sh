#!/bin/bash
# quick A/B: gpu tests (fail fast) + bench with roofline, no cpu baseline
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_quick.json'))
print('value %.1fM env-steps/s  ms/step %.3f'%(d['value']/1e6, d['ms_per_step']))
for k,v in d['roofline']['kernels'].items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items()})
PY
tail -3 gpurun_out/bench_quick.err
