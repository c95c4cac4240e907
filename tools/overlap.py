"""developer tool: how much of the staged host->device slab traffic runs under compute kernels.
usage: python tools/overlap.py <rocprofv3 output dir with kernel_trace.csv and memory_copy_trace.csv>
The copy trace carries no sizes: copies are selected by duration (>= 20 us: the slab arrays, 0.6 - 13 MB each; the work
tables and scalars are microseconds)."""
import csv, glob, sys
d = sys.argv[1]
kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
mt = glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True)[0]
kern = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(kt))
              if r['Kernel_Name'].lstrip('void ').startswith('k_'))
# merge kernel intervals
merged = []
for s, e in kern:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
rows = list(csv.DictReader(open(mt)))
tot = ov = n = 0
import bisect
starts = [m[0] for m in merged]
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e - s < 20000 or 'HOST_TO_DEVICE' not in r['Direction']:
        continue
    tot += e - s; n += 1
    i = max(bisect.bisect_right(starts, s) - 1, 0)
    while i < len(merged) and merged[i][0] < e:
        ov += max(0, min(e, merged[i][1]) - max(s, merged[i][0]))
        i += 1
span = (kern[-1][1] - kern[0][0]) / 1e6
print('host->device slab copies (>= 20 us each): %d copies, %.3f ms of copy time in a %.1f ms trace' % (n, tot / 1e6, span))
print('  of that time under a compute kernel of the library: %.3f ms = %.1f %%' % (ov / 1e6, 100.0 * ov / max(tot, 1)))
