#!/bin/bash
# developer tool: the PMC passes of tools/gpu_pmc.sh on BASELINE config 4 (Ant shapes, cooperative kernels) plus the two HBM
# counters -> gpurun_out/pmc4/summary.txt (copied to profiles/rNN_pmc_config4.txt)
mkdir -p gpurun_out/pmc4
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/gpurun_out/pmc4/p$i -o p$i -- python $ROOT/bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-plugin-path > /dev/null 2> $ROOT/gpurun_out/pmc4/p$i.err
  echo "pmc pass $i rc=$?"
done
cd $ROOT
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc4/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/pmc4/summary.txt', 'w') as out:
    out.write('BASELINE config 4 (40 tasks x 20 paths x 200 steps, obs 111, act 8, 2x128 MLP): per-dispatch means of separate --pmc passes\n')
    out.write('(FETCH_SIZE / WRITE_SIZE in KiB as reported; HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB, MI355X_MICROARCH.md)\n')
    for k, d in sorted(acc.items()):
        if not any(s in k for s in ('wide', 'k_wb', 'k_gae', 'k_returns', 'k_normalize', 'k_reduce')): continue
        out.write(k + '\n')
        for c, v in sorted(d.items()):
            out.write('   %-28s mean %16.1f  n %d\n' % (c, sum(v) / len(v), len(v)))
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            f, w = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']), sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE'])
            out.write('   %-28s %.1f MB per launch\n' % ('HBM traffic', (2 * f + w) * 1024 / 1e6))
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d:
            b, g = sum(d['SQ_VALU_MFMA_BUSY_CYCLES']) / len(d['SQ_VALU_MFMA_BUSY_CYCLES']), sum(d['GRBM_GUI_ACTIVE']) / len(d['GRBM_GUI_ACTIVE'])
            out.write('   %-28s %.3f  (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs))\n' % ('matrix pipe busy', b / (1024.0 * g / 8.0)))
        if 'SQ_ACTIVE_INST_VALU' in d and 'SQ_WAVE_CYCLES' in d:
            out.write('   %-28s %.3f\n' % ('VALU-active / wave cycles', (sum(d['SQ_ACTIVE_INST_VALU']) / len(d['SQ_ACTIVE_INST_VALU'])) / (sum(d['SQ_WAVE_CYCLES']) / len(d['SQ_WAVE_CYCLES']))))
print(open('gpurun_out/pmc4/summary.txt').read())
PY
rm -rf gpurun_out/pmc4/p*/
