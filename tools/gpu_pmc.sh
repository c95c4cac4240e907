#!/bin/bash
# quick bench + PMC counter passes (separate rocprofv3 runs, kernel-trace only)
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_quick.json'))
print('value %.1fM env-steps/s  ms/step %.3f'%(d['value']/1e6, d['ms_per_step']))
for k,v in d['roofline']['kernels'].items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items()})
PY
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/pmc/counters_list.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/p$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc/p$i.err
  echo "pmc pass $i rc=$?"
done
ls -R $GRAFT_REPO_ROOT/gpurun_out/pmc | head -40
