"""Oracle for the device-side exploration noise (TEST INFRASTRUCTURE ONLY).

Philox4x32-10 (J. Salmon, M. Moraes, R. Dror, D. Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the
Random123 library) restated in NumPy integer arithmetic, plus the Box-Muller step of promp_kernels_rollout.h.  The
reference draws its exploration noise with NumPy's global RNG on the host (policies/meta_gaussian_mlp_policy.py:131-134),
so there is no reference stream to match: what is pinned is (a) this restatement against the published known-answer
vectors of the generator (tests/test_oracle_philox.py) and (b) the device kernels against this restatement."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(counter, key):
    """counter [..., 4] uint32, key [..., 2] uint32 -> [..., 4] uint32"""
    c = np.array(counter, dtype=np.uint32, copy=True)
    k = np.array(np.broadcast_to(np.asarray(key, dtype=np.uint32), c.shape[:-1] + (2,)), copy=True)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c[..., 0].astype(np.uint64)
            p1 = M1 * c[..., 2].astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[..., 1] ^ k[..., 0]
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[..., 3] ^ k[..., 1]
            n3 = p0.astype(np.uint32)
            c = np.stack([n0, n1, n2, n3], axis=-1)
            k = np.stack([k[..., 0] + W0, k[..., 1] + W1], axis=-1)
    return c


def box_muller(a, b):
    """two uint32 words -> two standard normals (float32 arithmetic as on the device)"""
    f = np.float32
    u1 = ((a >> np.uint32(8)).astype(f) + f(1.0)) * f(1.0 / 16777216.0)
    u2 = (b >> np.uint32(8)).astype(f) * f(1.0 / 16777216.0)
    r = np.sqrt(f(-2.0) * np.log(u1))
    ang = f(6.283185307179586) * u2
    return (r * np.cos(ang)).astype(f), (r * np.sin(ang)).astype(f)


def action_noise(seed, rows, act_dim, stream):
    """[len(rows), act_dim] standard normals: row -> counter (row lo, row hi, action pair, stream), key = seed"""
    rows = np.asarray(rows, dtype=np.uint64)
    out = np.zeros((rows.size, act_dim), np.float32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    for pair in range((act_dim + 1) // 2):
        ctr = np.stack([(rows & np.uint64(0xFFFFFFFF)).astype(np.uint32), (rows >> np.uint64(32)).astype(np.uint32),
                        np.full(rows.size, pair, np.uint32), np.full(rows.size, stream, np.uint32)], axis=-1)
        r = philox4x32_10(ctr, key)
        n0, n1 = box_muller(r[:, 0], r[:, 1])
        out[:, 2 * pair] = n0
        if 2 * pair + 1 < act_dim:
            out[:, 2 * pair + 1] = n1
    return out
