"""Developer tool: cycle stamps of workgroup 0 / thread 0 inside k_pass / k_chain_hvp on config-3 shapes.
Needs a library built with -DPROMP_DEV_STAMPS (tools/build_variant.sh stamps -DPROMP_DEV_STAMPS; copy it over
promp_amd/libpromp_hip.so for the run)."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, '.')
from promp_amd import _lib, synthetic

M, P, T, O, A, hidden = 40, 20, 200, 20, 6, (64, 64)
rng = np.random.RandomState(0)
theta = synthetic.init_theta(rng, O, hidden, A)
ctx = _lib.Context(M, O, A, hidden, 1, max_rows=M * P * T, max_paths=M * P)
ctx.set_theta(theta); ctx.set_step_sizes(np.full(ctx.n_params, 0.1, np.float32)); ctx.switch_to_pre_update()
p0 = synthetic.make_paths(rng, theta, M, P, T, O, A, hidden)
f0 = _lib.flatten_paths(p0)
ctx.upload_step(0, f0['task_path_offsets'], f0['path_row_offsets'], f0['obs'], f0['rew'], f0['act'], f0['old_mean'], f0['old_log_std'])
ctx.process_samples(0, normalize_adv=True)
fn = ctx.lib.cdll.promp_debug_phase_stamps
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
names_hvp = ['L1', 'L2', 'L3', 'epi', 'dW3', 'dz2', 'dW2', 'dz1', 'dW1']
names_pass = ['0:X,L1', '1:tanh,split,L2own..B1', '2:L2oth', '3:tanh,split', '4:L3..B2', '5:epi', '6:dH2,dW3', '7:dz2,dW2,dH1own..B3', '8:dH1oth', '9:dz1,dW1', '-']  # k_pass_pair
import os
for hvp in tuple(int(x) for x in os.environ.get('PROMP_STAMP_KERNELS', '0,1').split(',')):
    for rep in range(3):
        buf = np.zeros(256 + 4096, np.uint64)
        rc = fn(ctx._h, 0, hvp, buf.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert rc == 0, ctx.lib.cdll.promp_last_error()
    s = buf.astype(np.int64)
    t0 = s[0]
    print('kernel', ('pass', 'hvp', 'hvp<CACHED>')[hvp], ' net1@%d nets@%d zeroed@%d' % (s[5] - t0, s[6] - t0, s[7] - t0), ' staged@%d  loop_end@%d  partial_written@%d  task_reduce_done@%d' % (s[1] - t0, s[2] - t0, s[3] - t0, s[4] - t0))
    wg = s[256:].reshape(-1, 4)
    wg = wg[wg[:, 0] > 0]
    w0 = wg[:, 0].min()
    end = np.maximum(wg[:, 1], wg[:, 2])
    two = wg[:, 2] > 0
    print('  workgroups %d (two-segment %d): start spread %.1f us; end min/median/max %.1f / %.1f / %.1f us after the first start; two-seg median end %.1f; one-seg median end %.1f'
          % (len(wg), two.sum(), (wg[:, 0].max() - w0) / 100.0, (end.min() - w0) / 100.0, np.median(end - w0) / 100.0, (end.max() - w0) / 100.0,
             np.median((end - w0)[two]) / 100.0 if two.any() else 0, np.median((end - w0)[~two]) / 100.0))
    print('  own duration (us) one-seg median %.1f max %.1f; two-seg median %.1f max %.1f' % (np.median((end - wg[:, 0])[~two]) / 100., ((end - wg[:, 0])[~two]).max() / 100., np.median((end - wg[:, 0])[two]) / 100. if two.any() else 0, ((end - wg[:, 0])[two]).max() / 100. if two.any() else 0))
    for tix in range(4):
        st = s[8 + 16 * tix: 8 + 16 * tix + 12]
        if st[0] == 0:
            continue
        names = names_hvp if hvp else names_pass
        nph = 9 if hvp else 10
        print('  tile', tix, 'start@%d' % (st[0] - t0), ' '.join('%s %d' % (names[i], st[i + 1] - st[i]) for i in range(nph)), ' total', st[nph] - st[0])
