"""Developer study (CPU, NumPy): accuracy of an FP32 GEMM emulated on BF16 matrix cores by error-compensated splitting.
x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1); products accumulated in float32 (what
v_mfma_f32_16x16x16_bf16 does: exact bf16 x bf16 products, float32 accumulation).  Reports the error of the 3-, 6- and
9-product variants against float64, next to plain float32, on a layer of the policy network's shape (tanh activations x
N(0, 1/sqrt(K)) weights, K = 64) and on a gradient-like contraction over 16 samples with small cotangents."""
import numpy as np


def bf16(x):
    """round-to-nearest-even float32 -> bfloat16, returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    x0 = bf16(x)
    r1 = (x - x0).astype(np.float32)
    x1 = bf16(r1)
    x2 = bf16((r1 - x1).astype(np.float32))
    return x0, x1, x2


def mm32(a, b):
    """float32 accumulation in k order (one fused chain per output like an MFMA accumulator)"""
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(a.shape[1]):
        acc = (acc + (a[:, k:k + 1].astype(np.float64) * b[k:k + 1, :].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return acc


def emulate(a, b, terms):
    A, B = split3(a), split3(b)
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for (i, j) in terms:                      # smallest terms first
        acc = (acc + mm32(A[i], B[j])).astype(np.float32)
    return acc


T3 = [(1, 0), (0, 1), (0, 0)]
T6 = [(1, 1), (2, 0), (0, 2)] + T3
T9 = [(2, 2), (2, 1), (1, 2)] + T6


def report(name, a, b):
    ref = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(ref).max()
    rows = [('float32 chain', mm32(a, b)), ('bf16 x1 (plain bf16)', emulate(a, b, [(0, 0)])), ('bf16 x3', emulate(a, b, T3)),
            ('bf16 x6', emulate(a, b, T6)), ('bf16 x9', emulate(a, b, T9))]
    print(name)
    for label, got in rows:
        err = np.abs(got.astype(np.float64) - ref)
        print('  %-22s max |err| / max |ref| = %.2e   rms = %.2e' % (label, err.max() / scale, np.sqrt((err ** 2).mean()) / scale))


rng = np.random.RandomState(0)
h = np.tanh(rng.randn(256, 64)).astype(np.float32)
w = (rng.randn(64, 64) / 8.0).astype(np.float32)
report('forward layer: tanh activations [256 x 64] x weights [64 x 64]', h, w)
dz = (rng.randn(16, 64) * 1e-4 * rng.rand(16, 1)).astype(np.float32)
report('weight gradient: activations^T [64 x 16] x cotangents [16 x 64]', np.ascontiguousarray(h[:16].T), dz)
