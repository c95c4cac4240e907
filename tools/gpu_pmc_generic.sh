#!/bin/bash
# developer tool: PMC passes over the (256,256) case of tools/generic_timing.py (layer-by-layer kernels) -> gpurun_out/pmcg/summary.txt
mkdir -p gpurun_out/pmcg
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES" "TA_BUSY_sum TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/gpurun_out/pmcg/p$i -o p$i -- python $ROOT/tools/generic_timing.py --steps 1 --case ${1:-1} > /dev/null 2> $ROOT/gpurun_out/pmcg/p$i.err
  echo "pmc pass $i ($set) rc=$?"
done
cd $ROOT
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcg/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/pmcg/summary.txt', 'w') as out:
    out.write('(256,256) policy, 40 tasks x 20 paths x 100 steps: per-dispatch means of separate --pmc passes (a kind mixes its layers)\n')
    out.write('(FETCH_SIZE / WRITE_SIZE in KiB as reported; HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB, MI355X_MICROARCH.md)\n')
    for k, d in sorted(acc.items()):
        if not any(s in k for s in ('k_gb', 'k_gen')): continue
        out.write(k + '\n')
        for c, v in sorted(d.items()):
            out.write('   %-32s mean %16.1f  n %d\n' % (c, sum(v) / len(v), len(v)))
        m = lambda n: sum(d[n]) / len(d[n])
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d: out.write('   %-32s %.1f MB per launch\n' % ('HBM traffic', (2 * m('FETCH_SIZE') + m('WRITE_SIZE')) * 1024 / 1e6))
        if 'TCC_HIT_sum' in d and 'TCC_REQ_sum' in d: out.write('   %-32s %.3f\n' % ('L2 hit rate', m('TCC_HIT_sum') / max(1.0, m('TCC_HIT_sum') + m('TCC_MISS_sum'))))
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d: out.write('   %-32s %.3f\n' % ('matrix pipe busy', m('SQ_VALU_MFMA_BUSY_CYCLES') / (1024.0 * m('GRBM_GUI_ACTIVE') / 8.0)))
print(open('gpurun_out/pmcg/summary.txt').read())
PY
rm -rf gpurun_out/pmcg/p*/
