#!/bin/bash
# developer tool: PMC counters of the pass kernels for every library variant under tools/ablate -> gpurun_out/pmc_ab.txt
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cp $ROOT/promp_amd/libpromp_hip.so /tmp/lib_keep.so
: > $ROOT/gpurun_out/pmc_ab.txt
for f in $ROOT/tools/ablate/lib_*.so; do
cp $f $ROOT/promp_amd/libpromp_hip.so
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/gpurun_out/pmc/p$i -o p$i -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> $ROOT/gpurun_out/pmc/p$i.err
done
cd $ROOT
echo "== $f" >> gpurun_out/pmc_ab.txt
python - >> gpurun_out/pmc_ab.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'k_pass' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s mean %14.1f  n %d' % (c, sum(v) / len(v), len(v)))
PY
rm -rf gpurun_out/pmc/p*/
done
cp /tmp/lib_keep.so $ROOT/promp_amd/libpromp_hip.so
cat $ROOT/gpurun_out/pmc_ab.txt
