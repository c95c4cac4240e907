"""Copy the judged summaries of gpurun_out/round/ into profiles/ (tracked) under a round tag, and derive the per-launch
HBM traffic of the pass kernels from the PMC passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 64 B per 128-B
request on gfx950 for wide coalesced reads => doubled; WRITE_SIZE used as reported; both in KiB)."""
import json
import os
import shutil
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src, dst = os.path.join(ROOT, 'gpurun_out/round'), os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)
for name, out in [('bench.json', '%s_bench.json'), ('pytest_gpu.log', '%s_pytest_gpu.log'), ('smoke.log', '%s_smoke.log'),
                  ('device.txt', '%s_device.txt'), ('trace/trace_kernel_stats.csv', '%s_rocprofv3_kernel_stats.csv'),
                  ('trace/trace_domain_stats.csv', '%s_rocprofv3_domain_stats.csv'),
                  ('bench_config4.json', '%s_bench_config4.json'), ('bench_config5.json', '%s_bench_config5.json'),
                  ('shard_timings.txt', '%s_shard_timings.txt'), ('shard_timings_config4.txt', '%s_shard_timings_config4.txt'), ('primal_cache_ab.txt', '%s_primal_cache_ab.txt'), ('h2d_overlap.txt', '%s_h2d_overlap.txt'), ('bench_staged.json', '%s_bench_staged_uploads.json'), ('timeline.txt', '%s_step_timeline.txt'), ('timeline_config4.txt', '%s_step_timeline_config4.txt'), ('generic_shapes.txt', '%s_generic_shapes.txt'), ('stage_a.txt', '%s_stage_a.txt'),
                  ('trace4/trace4_kernel_stats.csv', '%s_rocprofv3_kernel_stats_config4.csv'),
                  ('full_size_parity.txt', '%s_full_size_parity.txt'), ('split_accuracy.txt', '%s_split_accuracy.txt'),
                  ('bench_config4_fp32_kernels.json', '%s_bench_config4_fp32_kernels.json'), ('wb_stamps.txt', '%s_wb_kernels_phase_stamps.txt'),
                  ('pmc_config4.txt', '%s_pmc_config4.txt')]:
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, out % tag))
def short_name(k):
    """k_pass: every gradient-pass launch of a step under one name (the mean over dispatches then weighs the launches that
    also fill the primal cache by their share, which is what bench.py's per-launch average does), the forward-only instance
    apart; the cache-filling launches additionally on their own"""
    base = k.split('(')[0].replace('void ', '')
    name, _, targs = base.partition('<')
    targs = [t.strip() for t in targs.rstrip('>').split(',')] if targs else []
    if name == 'k_pass' and len(targs) >= 4 and targs[3] == 'false':
        return 'k_pass<fwd-only>'
    return name


rows = {}
for i in range(1, 9):
    f = os.path.join(src, 'pmc%d/pmc%d_counter_collection.csv' % (i, i))
    if not os.path.exists(f):
        continue
    df = pd.read_csv(f)
    df['Kernel_Name'] = df['Kernel_Name'].map(short_name)
    g = df.groupby(['Kernel_Name', 'Counter_Name'])['Counter_Value'].mean().unstack()
    for k, r in g.iterrows():
        rows.setdefault(k, {}).update({c: float(v) for c, v in r.items()})
pmc = pd.DataFrame(rows).T
pmc.index.name = 'kernel'
pmc.to_csv(os.path.join(dst, '%s_rocprofv3_pmc_per_dispatch_mean.csv' % tag))
traffic = {}
for k, r in pmc.iterrows():
    if 'FETCH_SIZE' in r and 'WRITE_SIZE' in r and r['FETCH_SIZE'] == r['FETCH_SIZE']:
        short = k
        traffic[short] = dict(fetch_size_kib=r['FETCH_SIZE'], write_size_kib=r['WRITE_SIZE'],
                              hbm_bytes_per_launch=(2.0 * r['FETCH_SIZE'] + r['WRITE_SIZE']) * 1024.0,
                              note='(2 x FETCH_SIZE + WRITE_SIZE) KiB; separate --pmc passes of bench.py --steps 3; gfx950 FETCH_SIZE x2 correction')
        # matrix-pipe utilisation: busy cycles summed over the chip's 1024 SIMDs against the kernel's own cycles (GRBM_GUI_ACTIVE is
        # summed over the 8 XCDs: k_pass 1.19 M per dispatch = 8 x the 148 k cycles of a 58 us launch)
        if r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) == r.get('SQ_VALU_MFMA_BUSY_CYCLES', float('nan')) and r.get('GRBM_GUI_ACTIVE', 0) > 0:
            traffic[short]['mfma_busy_frac'] = float(r['SQ_VALU_MFMA_BUSY_CYCLES']) / (1024.0 * float(r['GRBM_GUI_ACTIVE']) / 8.0)
            traffic[short]['valu_active_frac'] = float(r.get('SQ_ACTIVE_INST_VALU', float('nan'))) / float(r['SQ_WAVE_CYCLES']) if r.get('SQ_WAVE_CYCLES', 0) > 0 else None
sys.path.insert(0, ROOT)
import bench
traffic['_kernel_sources_sha256'] = bench.kernel_sources_sha256()    # bench.py reports these counters only for the same sources
json.dump(traffic, open(os.path.join(dst, '%s_hbm_traffic.json' % tag), 'w'), indent=1)
print(pmc.loc[[k for k in pmc.index if 'k_' in k]].T.to_string())
print(json.dumps({k: round(v['hbm_bytes_per_launch'] / 1e6, 2) for k, v in traffic.items() if not k.startswith('_')}, indent=1))
