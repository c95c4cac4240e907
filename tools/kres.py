"""Developer tool: compile the library for gfx950 and print register / spill figures of selected kernels.
usage: python tools/kres.py [substring ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pats = sys.argv[1:] or ['k_chain']
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize',
       '-shared', '-fPIC', os.path.join(ROOT, 'promp_amd/csrc/promp_hip.hip'), '-o', os.path.join(ROOT, 'promp_amd/libpromp_hip.so'),
       '-lrccl', '-Wno-pass-failed', '-Rpass-analysis=kernel-resource-usage'] + os.environ.get('PROMP_EXTRA_FLAGS', '').split()
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.split('\n'):
    m = re.search(r'remark:\s+(.*?): (.*) \[-Rpass', line)
    if 'error' in line: print(line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == 'Function Name':
        cur = v; rows[cur] = {}
    elif cur: rows[cur][k] = v
for name, r in rows.items():
    if any(p in name for p in pats):
        print('%-60s VGPR %s AGPR %s spillV %s spillS %s occ %s' % (name, r.get('VGPRs'), r.get('AGPRs'), r.get('VGPRs Spill'), r.get('SGPRs Spill'), r.get('Occupancy [waves/SIMD]')))
