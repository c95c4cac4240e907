"""developer tool: per-launch timeline of the last bench step from a rocprofv3 kernel trace csv (name, duration, gap to the
previous kernel's end) -- shows where a step's wall time goes when the kernels are small (the 5-task shard)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# a step starts at the first k_returns after a k_final_adam / k_mean_adam
starts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_returns') or 'k_returns' in r['Kernel_Name']]
steps = [i for j, i in enumerate(starts) if j % 2 == 0]
import os
# default: a step from the middle of the timed loop of the 10-step runs gpu_trace.sh / gpu_round.sh trace (the list also
# holds the set-up call, the warm-up steps and, after the timed loop, the upload / staged-upload measurements)
sel = int(os.environ.get('TIMELINE_STEP', '6'))
a, b = steps[sel], steps[sel + 1]
t0 = int(rows[a]['Start_Timestamp']); prev = None; busy = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = 0 if prev is None else s - prev
    busy += e - s
    print('%8.1f us  dur %7.1f  gap %6.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, r['Kernel_Name'][:60]))
    prev = e
print('step wall %.1f us, kernel busy %.1f us, %d launches' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, busy / 1e3, b - a))
