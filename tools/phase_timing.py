"""Developer tool: cycle stamps of workgroup 0 / lane 0 inside k_fwd_bwd (and k_hvp) on config-3 shapes."""
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
for hvp in (0, 1):
    for rep in range(3):
        buf = np.zeros(256, np.uint64)
        rc = fn(ctx._h, 0, hvp, buf.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert rc == 0, ctx.lib.cdll.promp_last_error()
    s = buf.astype(np.int64)
    t0 = s[0]
    print('kernel', 'hvp' if hvp else 'fwd_bwd', ' stage->%d  loop_end->%d  reduce_end->%d  write_end->%d' % (s[1] - t0, s[2] - t0, s[3] - t0, s[4] - t0))
    if not hvp:
        print('  end phase (cycles after loop end): ', [int(x - s[2]) for x in s[120:128]], ' write_end', int(s[4] - s[2]))
    for tix in range(4):
        st = s[8 + 16 * tix: 8 + 16 * tix + 16]
        if st[0] == 0:
            continue
        sub = st[9:13] - st[0]
        st = st[:9]
        nz = [int(x) for x in st if x > 0]
        print('  tile', tix, 'start@%d' % (nz[0] - t0), 'phase deltas:', [nz[i + 1] - nz[i] for i in range(len(nz) - 1)], ' sub(after Xstore, after loads issued, after L1 gemm, after L2 gemm):', [int(x) for x in sub])
