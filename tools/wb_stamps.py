"""Developer tool: cycle stamps of workgroup 0 / thread 0 inside the BF16-pipe cooperative kernels (promp_kernels_wide_bf16.h) on
config-4 shapes.  Needs a library built with -DPROMP_DEV_STAMPS (tools/build_variant.sh stamps -DPROMP_DEV_STAMPS)."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, '.')
from promp_amd import _lib, synthetic

M, P, T, O, A, hidden = 40, 20, 200, 111, 8, (128, 128)
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
names = {0: ['B0 (top barrier)', 'row-data request, L1, park', 'xput, xreq', 'B2', 'L2', 'L3 partial, ring begin', 'B3', 'epilogue', 'B4',
             'dW3, dH2, dz2, db2', 'B5', 'dH1 + dW2 fused, dz1', '-', 'dW1'],
         1: ['B0, xput, xreq, row-data request', 'B1', 'F1 (layer 1 + tangent)', 'B2', 'F2 (layer 2 + tangent)', 'F3 (partial means)', 'B3',
             'loss-level R-operator', 'B4', 'dW3, dH2, ndz2, qz2, db2', 'B5', 'backward chains + dW2', 'B6', 'qz1 store, dW1']}
for hvp in (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '0').split(',')):
    for rep in range(3):
        buf = np.zeros(256 + 4096, np.uint64)
        rc = fn(ctx._h, 0, hvp, buf.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert rc == 0, ctx.lib.cdll.promp_last_error()
    s = buf.astype(np.int64)[8:8 + 40]
    nm = names[hvp]
    n = len(nm)
    print('kernel', ('k_wb_fwd_bwd', 'k_wb_hvp')[hvp], ' second round of workgroup 0 (thread 0): %d cycles' % (s[n] - s[0]))
    for i in range(n):
        print('   %-34s %7d' % (nm[i], s[i + 1] - s[i]))
    print('   finer stamps (cycles after the round\'s first):', {i: int(s[i] - s[0]) for i in range(17, 40) if s[i]})
