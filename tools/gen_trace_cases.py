"""Developer tool: per-case kernel breakdown of a rocprofv3 kernel trace of tools/generic_timing.py.
usage: python tools/gen_trace_cases.py gpurun_out/generic/trace/gen_kernel_trace.csv"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
case = 0
stats = [collections.OrderedDict() for _ in range(4)]
for r in rows:
    n = r['Kernel_Name']
    if case == 0 and ', 4>' in n and 'k_g' in n: case = 1
    if case == 1 and 'k_gram_wide' in n: case = 2
    if case == 2 and 'k_pass<' in n: case = 3
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    key = n.replace('void ', '').split('(')[0] + ' grid ' + str(int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) + 'x' + r['Grid_Size_Y']
    s = stats[case].setdefault(key, [0, 0.0]); s[0] += 1; s[1] += d
for c in range(3):
    print('--- case', c)
    tot = sum(v[1] for v in stats[c].values())
    for k, v in sorted(stats[c].items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 18]:
        print('%-60s calls %5d avg %9.1f us total %8.2f ms %5.1f%%' % (k[:60], v[0], v[1] / v[0], v[1] / 1e3, 100 * v[1] / tot))
