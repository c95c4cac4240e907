#!/bin/bash
# developer tool: PMC counter passes (separate rocprofv3 runs, kernel-trace only) -> gpurun_out/pmc/summary.txt
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/gpurun_out/pmc/p$i -o p$i -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> $ROOT/gpurun_out/pmc/p$i.err
  echo "pmc pass $i rc=$?"
done
cd $ROOT
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/pmc/summary.txt', 'w') as out:
    for k, d in acc.items():
        if 'chain' not in k and 'wide' not in k and 'k_pass' not in k: continue
        out.write(k + '\n')
        for c, v in sorted(d.items()):
            out.write('   %-28s mean %14.1f  n %d\n' % (c, sum(v) / len(v), len(v)))
print(open('gpurun_out/pmc/summary.txt').read())
PY
rm -rf gpurun_out/pmc/p*/
