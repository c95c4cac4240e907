cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "hvp or primal_cache or meta or constraint or split" 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > /tmp/b.json; python - <<'PY'
import json
d=json.load(open('/tmp/b.json')); print(d['ms_per_step'], d['value'], {k:round(v['avg_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})
PY
