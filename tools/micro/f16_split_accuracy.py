"""Developer study (CPU, NumPy): an FP32 GEMM on the FP16 matrix instructions by a TWO-term split, three products.

    x = hi + lo,  hi = f16(x),  lo = f16(x - hi)          (round to nearest even; the residual is exact in float32)
    a b ~ hi_a hi_b + hi_a lo_b + lo_a hi_b               (lo_a lo_b ~ 2^-22 |a b| dropped)

FP16 has 11 significant bits, so hi + lo carries 22 -- provided lo stays a normal number or the matrix instruction keeps
subnormal inputs (an f16 subnormal has an absolute error of 2^-25 whatever its size; flushed to zero it is 2^-14 2^-11).  The range
is the price: |x| <= 65504, and below 2^-14 hi itself is subnormal.  Reported: the error of the 2-term / 3-product form against
float64 next to the float32 chain and the BF16 3-term / 6-product form it would replace, on the operands of the policy passes --
weights N(0, 1/sqrt(K)), tanh activations, cotangents of size adv / N -- with and without a power-of-two prescale of the
cotangents, with lo flushed (ftz) or kept (subnormal), and with lo carried at 2^11 in a second accumulator."""
import numpy as np


def bf16(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split_bf16_3(x):
    x = np.asarray(x, np.float32)
    x0 = bf16(x)
    r1 = (x - x0).astype(np.float32)
    x1 = bf16(r1)
    return x0, x1, bf16((r1 - x1).astype(np.float32))


def f16(x, ftz=False):
    with np.errstate(over='ignore'):
        h = np.asarray(x, np.float32).astype(np.float16)
    if ftz:
        h = np.where(np.abs(h) < np.float16(6.1035e-5), np.float16(0), h)
    return h.astype(np.float32)


def split_f16_2(x, ftz=False, lo_scale=1.0):
    x = np.asarray(x, np.float32)
    hi = f16(x, ftz)
    lo = f16(((x - hi) * np.float32(lo_scale)).astype(np.float32), ftz)
    return hi, lo


def mm32(a, b):
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(a.shape[1]):
        acc = (acc + (a[:, k:k + 1].astype(np.float64) * b[k:k + 1, :].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return acc


def emu_bf16x6(a, b):
    A, B = split_bf16_3(a), split_bf16_3(b)
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for (i, j) in [(2, 0), (1, 1), (1, 0), (0, 2), (0, 1), (0, 0)]:
        acc = (acc + mm32(A[i], B[j])).astype(np.float32)
    return acc


def emu_f16x3(a, b, ftz=False, sa=1.0, sb=1.0):
    """one accumulator: lo.hi, hi.lo, hi.hi; operands prescaled by the powers of two sa, sb (undone on the result)"""
    A, B = split_f16_2(a * np.float32(sa), ftz), split_f16_2(b * np.float32(sb), ftz)
    acc = mm32(A[1], B[0])
    acc = (acc + mm32(A[0], B[1])).astype(np.float32)
    acc = (acc + mm32(A[0], B[0])).astype(np.float32)
    return (acc / np.float32(sa * sb)).astype(np.float32)


def emu_f16x3_two_acc(a, b, sa=1.0, sb=1.0):
    """lo carried at 2^11 (always normal), cross products in a second accumulator merged by an exact 2^-11"""
    A, B = split_f16_2(a * np.float32(sa), True, 2048.0), split_f16_2(b * np.float32(sb), True, 2048.0)
    x = (mm32(A[1], B[0]) + mm32(A[0], B[1])).astype(np.float32)
    acc = (mm32(A[0], B[0]) + x * np.float32(1.0 / 2048.0)).astype(np.float32)
    return (acc / np.float32(sa * sb)).astype(np.float32)


def report(name, a, b, sa=1.0, sb=1.0):
    ref = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(ref).max()
    rows = [('float32 chain', mm32(a, b)), ('bf16 3-term x6', emu_bf16x6(a, b)),
            ('f16 2-term x3, subnormals kept', emu_f16x3(a, b, False, sa, sb)),
            ('f16 2-term x3, flushed', emu_f16x3(a, b, True, sa, sb)),
            ('f16 2-term x3, lo at 2^11, 2 acc', emu_f16x3_two_acc(a, b, sa, sb))]
    print('%s   (prescale a %g, b %g)' % (name, sa, sb))
    for label, got in rows:
        err = np.abs(got.astype(np.float64) - ref)
        print('  %-34s max |err| / max |ref| = %.2e   rms = %.2e' % (label, err.max() / scale, np.sqrt((err ** 2).mean()) / scale))


if __name__ == '__main__':
    rng = np.random.RandomState(0)
    x = rng.randn(256, 32).astype(np.float32)
    x[:, 20:] = 0
    w1 = (rng.randn(32, 64) / np.sqrt(20.0) * 2.885).astype(np.float32)
    report('layer 1: observations [256 x 32] x W1 [32 x 64] (tanh prescale folded)', x, w1)
    h = np.tanh(rng.randn(256, 64)).astype(np.float32)
    w = (rng.randn(64, 64) / 8.0 * 2.885).astype(np.float32)
    report('layer 2: tanh activations [256 x 64] x W2 [64 x 64]', h, w)
    w3 = (rng.randn(64, 6) / 8.0).astype(np.float32)
    report('output layer: [256 x 64] x W3 [64 x 6]', h, w3)
    N = 4000.0
    dz = (rng.randn(16, 64) * rng.randn(16, 1) / N).astype(np.float32)
    for s in (1.0, 4096.0):
        report('weight gradient: H^T [64 x 16] x dZ [16 x 64], dZ ~ adv / N', np.ascontiguousarray(h[:16].T), dz, 1.0, s)
        report('backward chain: W2 [64 x 64] x dZ2^T [64 x 16]', (w / 2.885).astype(np.float32), np.ascontiguousarray(dz.T), 1.0, s)
    dmu = (rng.randn(16, 6) * rng.randn(16, 1) / N).astype(np.float32)
    report('output-kernel gradient: H2^T [64 x 16] x dmu [16 x 6]', np.ascontiguousarray(h[:16].T), dmu, 1.0, 4096.0)
    xb = (x * 37.0).astype(np.float32)
    report('layer 1 with observations of size 37', xb, w1)
