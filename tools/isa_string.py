"""Developer tool: the instruction stream of a kernel in a hipcc -S listing as a string of one character per instruction
(M / W: BF16 MFMA 16x16x32 / 32x32x16, F: FP32 MFMA, v: vector ALU, c: v_cvt_pk, a: v_accvgpr_*, r / w: LDS read / write,
g: global, S: scratch, |: s_waitcnt, n: s_nop, s: other scalar), from 120 instructions in front of the first MFMA to 40 behind
the last.  Shows at a glance where the matrix blocks are dense, where waits sit in front of products, where copies pile up.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only promp_amd/csrc/promp_hip.hip -o /tmp/promp.s
       python tools/isa_string.py /tmp/promp.s _Z11k_chain_hvpILi4ELi4ELi5ELi4ELb1EEv8PassArgs"""
import collections
import re
import sys

path, name = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
s0 = [i for i, l in enumerate(lines) if l.startswith(name + ':')][0]
e0 = [i for i in range(s0, len(lines)) if 's_endpgm' in lines[i]][0]
seg = lines[s0:e0]
idx = [i for i, l in enumerate(seg) if 'v_mfma' in l]
print('instructions of the kernel: %d; MFMAs between %d and %d; scratch accesses at %s'
      % (len(seg), idx[0], idx[-1], [i for i, l in enumerate(seg) if 'scratch_' in l]))
seg = seg[max(0, idx[0] - 120):idx[-1] + 40]


def cat(l):
    m = re.match(r'\s+([a-z_0-9]+)', l)
    if not m:
        return None
    k = m.group(1)
    if k.startswith('v_mfma_f32_16x16x32'): return 'M'
    if k.startswith('v_mfma_f32_32x32x16'): return 'W'
    if k.startswith('v_mfma'): return 'F'
    if k.startswith('v_accvgpr'): return 'a'
    if k.startswith('ds_read'): return 'r'
    if k.startswith('ds_write'): return 'w'
    if k.startswith('global'): return 'g'
    if k.startswith('scratch'): return 'S'
    if k == 's_waitcnt': return '|'
    if k == 's_nop': return 'n'
    if k.startswith('s_'): return 's'
    if k.startswith('v_cvt_pk'): return 'c'
    if k.startswith('v_'): return 'v'
    return '?'


s = ''.join(c for c in map(cat, seg) if c)
print(len(s), dict(collections.Counter(s)))
for i in range(0, len(s), 150):
    print(s[i:i + 150])
