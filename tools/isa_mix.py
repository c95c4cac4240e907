"""Developer tool: instruction mix of the hottest loop of a kernel in a hipcc -S listing.
usage: python tools/isa_mix.py /tmp/promp.s _Z6k_passILi4ELi4ELi4ELb1ELb0EEv8PassArgs [--all]
Finds the loop (label .. backward branch) with the most MFMAs and prints counts per mnemonic class."""
import collections
import re
import sys


def kernel_lines(path, name):
    out, on = [], False
    for ln in open(path):
        if ln.startswith(name + ':'):
            on = True
            continue
        if on:
            if ln.strip().startswith('s_endpgm'):
                break
            s = ln.split(';')[0].rstrip()
            if s.strip():
                out.append(s)
    return out


def classify(m):
    if m.startswith('v_mfma'):
        return 'mfma'
    if m.startswith('ds_read') or m.startswith('ds_load'):
        return 'ds_read'
    if m.startswith('ds_write') or m.startswith('ds_store'):
        return 'ds_write'
    if m.startswith('ds_'):
        return 'ds_other'
    if m.startswith('global_') or m.startswith('buffer_') or m.startswith('flat_') or m.startswith('scratch_'):
        return 'vmem'
    if m in ('v_exp_f32_e32', 'v_rcp_f32_e32', 'v_log_f32_e32', 'v_rsq_f32_e32', 'v_sqrt_f32_e32', 'v_exp_f32', 'v_rcp_f32'):
        return 'trans'
    if m.startswith('v_accvgpr'):
        return 'accvgpr'
    if m.startswith('v_'):
        if m.startswith('v_pk_'):
            return 'valu_pk'
        if m.endswith('_e32'):
            return 'valu_e32'
        if m.endswith('_e64') or m.endswith('_dpp') or m.endswith('_sdwa'):
            return 'valu_e64'
        return 'valu_vop3'
    if m.startswith('s_waitcnt'):
        return 's_waitcnt'
    if m.startswith('s_nop'):
        return 's_nop'
    if m.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    path, name = sys.argv[1], sys.argv[2]
    L = kernel_lines(path, name)
    labels = {}
    for i, s in enumerate(L):
        mm = re.match(r'^(\.LBB\d+_\d+):', s)
        if mm:
            labels[mm.group(1)] = i
    loops = []
    for i, s in enumerate(L):
        t = s.split()
        if t and t[0].startswith('s_cbranch') or (t and t[0] == 's_branch'):
            tgt = t[-1]
            if tgt in labels and labels[tgt] < i:
                body = L[labels[tgt]:i + 1]
                nm = sum(1 for b in body if b.split()[0].startswith('v_mfma'))
                loops.append((nm, labels[tgt], i))
    loops.sort(reverse=True)
    if not loops:
        print('no loop')
        return
    for nm, a, b in loops[:(3 if '--all' in sys.argv else 1)]:
        body = [x for x in L[a:b + 1] if not x.lstrip().startswith('.')]
        cnt = collections.Counter(classify(x.split()[0]) for x in body)
        mn = collections.Counter(x.split()[0] for x in body)
        print('loop lines %d..%d: %d instructions' % (a, b, len(body)))
        for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
            print('  %-10s %5d' % (k, v))
        print('  top mnemonics:', ', '.join('%s %d' % kv for kv in mn.most_common(40)))


main()
