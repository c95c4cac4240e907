// promp_kernels_wide_bf16.h -- the policy passes for two hidden layers of 128 units on the 16-bit matrix pipe (round 5: three BF16
// terms, six products; round 6: two FP16 terms, three products with data-following scales -- promp_device.h: split_pair,
// DESIGN.md sections 5.1, 5.5; "BF16" in the comments below names round 5's format, the layouts are the same)
// (BASELINE config 4: AntRandDirec, obs 111, act 8, 2x128 tanh MLP; reference policies/networks/mlp.py:65-119,
// envs/mujoco_envs/ant_rand_direc.py:52-57; arithmetic: oracle/promp.py = meta_algos/pro_mp.py:59-155, base.py:192-215).
//
// Same mathematics, arguments, work-item table and partial-row layout as k_wide_fwd_bwd / k_wide_hvp
// (promp_kernels_policy_wide.h); what changes is the arithmetic pipe and who keeps what.  FP32 MFMA issues at the vector rate
// on gfx950 (1/16 of the BF16 rate), and the FP32 cooperative kernels had that pipe 62-69 % busy: the formulation was exhausted.
// Here every large GEMM runs as v_mfma_f32_32x32x16_bf16 in float32-equivalent arithmetic, k_pass's way: both operands split
// into three BF16 terms (x = t0 + t1 + t2 up to 2^-24 |x|), the six largest of the nine cross products, float32 accumulation.
//
//   * 4 waves, ONE per SIMD (512 registers each).  Wave w OWNS units 32 w .. 32 w + 31 of both hidden layers: it produces those
//     units of H1 / H2 / dZ2 / dZ1 and accumulates exactly those columns of the two hidden kernels' gradients (its 32 rows of the
//     output kernel) over all the rows the workgroup walks: every partial entry is written once, by one lane, in a fixed order.
//   * Activations are TRANSPOSED in the matrix instruction (units along m, samples along n), as in k_pass: the weights are the A
//     operand, the activations of a 32-row round are the B operand and live in LDS as three BF16 planes per tensor, sample-major:
//     one ds_read_b128 hands a lane the eight consecutive units of its sample that a K = 16 step contracts.  A D fragment (lane =
//     sample, four consecutive units per register quad) goes back to LDS with one ds_write_b64 per quad and plane.
//   * The weights never touch LDS.  k_wb_planes splits a parameter vector ONCE per pass into BF16 planes in the order the A
//     fragments are read (global memory, L2-resident: 276 KB per task); the first-order pass keeps its forward slices (hidden_0
//     columns 84, hidden_1 columns 96 registers) for the whole work item and streams the hidden_1 ROW slice of the backward product
//     (24 KB per wave and round) through a register ring a few K steps ahead of the products; the R-operator pass streams all six
//     slices (theta and the direction).
//   * Weight gradients contract over samples: both operands come back from the same tiles through ds_read_b64_tr_b16 (the
//     4 x 16 transpose read: two reads give a lane the eight samples of its unit), no second copy of anything.
//   * The 8-byte chunk (sample s, units 4 c .. 4 c + 3) sits at chunk position c ^ sigma(s) of its 256-byte sample row, sigma a
//     bit permutation of s chosen so that all three access patterns are conflict-free: the 16 lanes of a ds_read_b128 service group
//     (16 samples, one 16-byte unit) hit 16 distinct 16-byte slots, the 32 lanes of a transpose read (4 samples x 8 chunks) all
//     64 banks once (the 16 writers of a chunk column meet two-way, which a ds_write_b64 absorbs).
//   * The bias gradient of hidden_0 rides in the hidden_0 kernel gradient for free: the observation tile carries a column of ones.
//   * The small products keep their natural shapes: action means on v_mfma_f32_16x16x32_bf16 (every wave contracts its own 32
//     units, the four partial sums are added in wave order by the distribution epilogue), the output-kernel gradient on the same
//     instruction over 32 samples (its padding row 15 collects the hidden_1 bias gradient from a row of ones), the cotangent
//     W3 dmu^T (K = act_dim <= 8) on the exact FP32 instruction.
//   * Nothing a round needs from memory is requested inside the round that uses it: the observations of round r + 1 are requested
//     during round r - 1 and split into the second observation tile during round r; the row data of the distribution epilogue
//     (actions, old means / log-stds, advantage: one (row, action) pair per thread) at the top of its round.
#pragma once
#include "promp_kernels_policy_wide.h"

// developer tooling: cycle stamps of workgroup 0 / thread 0 in its second round (-DPROMP_DEV_STAMPS, tools/wb_stamps.py)
#ifdef PROMP_DEV_STAMPS
#define WB_STAMP(i) do { if (a.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && rix == 1) a.dbg[8 + (i)] = promp_clock(); } while (0)
#else
#define WB_STAMP(i) do { } while (0)
#endif
#define WB_R 32            // rows per round
#define WB_ROWW 64         // words per sample row of a plane (128 bf16)
#define WB_PLANE 2048      // words per plane of a [32 samples][128 units] tile
#define WB_TILE (PROMP_NT * WB_PLANE)   // the planes of the split (round 6: two FP16 terms; -DPROMP_SPLIT_TERMS=3: three BF16 terms)
#define WB_MS 17           // row stride of the float32 mean / cotangent tile
#define WB_DMROW 8         // words per sample row of a cotangent plane (16 action slots)
#define WB_DMPL 256        // words per cotangent plane
// FP16 split (see promp_kernels_pass.h: pass_cotangent_scale, promp_kernels_chain.h: CHAIN_*): where the direction's largest entry,
// the first round's largest cotangent and -- when a work item is walked again -- its largest cotangent overall go
#ifndef PROMP_CHAIN_V_TARGET
#define PROMP_CHAIN_V_TARGET 3
#endif
#define WB_V_TARGET PROMP_CHAIN_V_TARGET

// ---- pre-split weight planes in global memory (k_wb_planes) ----
// per task:  C1 [w 4][q NKO][t 3][lane 64] x 16 B   hidden_0 kernel, column slices: lane (i, h) of (w, q): W1[16 q + 8 h + e][32 w + i]
//            C2 [w 4][q 8][t 3][lane 64] x 16 B     hidden_1 kernel, column slices:                         W2[16 q + 8 h + e][32 w + i]
//            R2 [w 4][q 8][t 3][lane 64] x 16 B     hidden_1 kernel, row slices (backward product):         W2[32 w + i][16 q + 8 h + e]
//            W3 [w 4][t 3][lane 64] x 16 B          output kernel, A operand of the 16x16x32 product: lane (a, g): W3[32 w + 8 g + e][a]
#define WB_FRAG (PROMP_NT * 256)          // words of one (wave, K step): its planes x 64 lanes x 16 B
PROMP_HD int wb_planes_c2(int nko) { return 4 * nko * WB_FRAG; }                  // word offsets inside a task's block
PROMP_HD int wb_planes_r2(int nko) { return 4 * nko * WB_FRAG + 4 * 8 * WB_FRAG; }
PROMP_HD int wb_planes_w3(int nko) { return 4 * nko * WB_FRAG + 2 * 4 * 8 * WB_FRAG; }
PROMP_HD int wb_planes_words(int nko) { return 4 * nko * WB_FRAG + 2 * 4 * 8 * WB_FRAG + 4 * WB_FRAG; }

struct WbPlaneArgs {
    const float* src;            // [tasks][Theta] (or one shared vector: src_stride 0, grid.y 1)
    long long src_stride;
    unsigned* dst;               // [tasks][wb_planes_words]
    int O, A, NKO;
    float row_sign;              // sign of the R2 region (the R-operator pass wants the direction's rows negated)
    // FP16 split: exact powers of two on the way into the planes --
    const float* obs_absmax;     // [tasks] or NULL: the hidden_0 kernel times the inverse of the observations' scale (wb_obs_shift)
    const float* vec_absmax;     // [tasks] or NULL: src is a direction of the R-operator pass; everything times its scale (wb_vec_scale)
};
PROMP_DEV int wb_obs_shift(const float* obs_absmax, int task) { return obs_shift(obs_absmax, task); }
// the direction's scale: its largest entry (k_vec_absmax) to [2^t, 2^(t+1)) (k_chain_hvp: chain_stage_nets)
PROMP_DEV float wb_vec_scale(const float* vec_absmax, int task, int t) {
    if (PROMP_NT != 2 || vec_absmax == nullptr) return 1.f;
    const float mx = vec_absmax[task];
    int k = (mx > 0.f && mx < 3.0e38f) ? scale_exp(mx, t) : 0;
    k = k < -100 ? -100 : k > 100 ? 100 : k;
    return pow2f(k);
}

struct LdsWB {
    int x0, x1, h1, h2, dz1, rh1, rh2, d1, mp, mp2, ms, ms2, dm, dm2, b1, b2, b3, vb1, vb2, vb3, ls, lmask, es, sn2, vls, red;
    int total;
};

PROMP_HD LdsWB wb_layout(bool hvp) {
    LdsWB L;
    int o = 0;
#define WB_TAKE(field, n) \
    L.field = o;          \
    o += ((n) + 3) & ~3
    WB_TAKE(x0, WB_TILE);
    WB_TAKE(h1, WB_TILE);
    WB_TAKE(h2, WB_TILE);
    L.x1 = L.dz1 = L.d1 = L.rh1 = L.rh2 = L.mp2 = L.ms2 = L.dm2 = L.vb1 = L.vb2 = L.vb3 = L.vls = 0;
    if (hvp) {
        WB_TAKE(rh1, WB_TILE);
        WB_TAKE(rh2, WB_TILE);
        WB_TAKE(d1, 4 * 1024);                  // [wave][quad 4][lane 64][4]: h1 of the wave's own units, parked between the passes
    } else {
        WB_TAKE(x1, WB_TILE);                   // second observation tile (round r + 1 is split while round r computes)
        WB_TAKE(dz1, WB_TILE);                  // dZ1, own columns only
        WB_TAKE(d1, 4 * 1024);                  // [wave][quad 4][lane 64][4]: 1 - h1^2 of the wave's own units, parked between the passes
    }
    WB_TAKE(mp, 4 * WB_R * 8);                  // [wave][sample][action]: partial means
    WB_TAKE(ms, WB_R * WB_MS);
    WB_TAKE(dm, PROMP_NT * WB_DMPL);
    if (hvp) {
        WB_TAKE(mp2, 4 * WB_R * 8);
        WB_TAKE(ms2, WB_R * WB_MS);
        WB_TAKE(dm2, PROMP_NT * WB_DMPL);
    }
    WB_TAKE(b1, 128);
    WB_TAKE(b2, 128);
    WB_TAKE(b3, 16);
    WB_TAKE(ls, 16);
    WB_TAKE(lmask, 16);
    WB_TAKE(es, 16);
    WB_TAKE(sn2, 16);
    WB_TAKE(red, 64);
    if (hvp) {
        WB_TAKE(vb1, 128);
        WB_TAKE(vb2, 128);
        WB_TAKE(vb3, 16);
        WB_TAKE(vls, 16);
    }
#undef WB_TAKE
    L.total = o;
    return L;
}

// chunk position swizzle: sample bits 0, 1 -> chunk bits 3, 4; sample bits 2, 3 -> chunk bits 1, 2 (even: 16-byte pairs stay pairs)
PROMP_DEV int wb_sigma(int s) { return ((s & 3) << 3) | (((s >> 2) & 3) << 1); }
// word offset, inside a plane, of the 8-byte chunk (sample s, units 4 c .. 4 c + 3)
PROMP_DEV int wb_chunk(int s, int c) { return s * WB_ROWW + 2 * (c ^ wb_sigma(s)); }

PROMP_DEV u32x4 wb_lds4(const float* p) { return *(const u32x4*)p; }

// B operand of a K = 16 step of v_mfma_f32_32x32x16_bf16: units 16 q + 8 h .. + 7 of sample j, three planes
PROMP_DEV void wb_read_b(u32x4 (&f)[PROMP_NT], const float* tile, int j, int h, int q) {
    const int off = wb_chunk(j, 4 * q + 2 * h);
#pragma unroll
    for (int t = 0; t < PROMP_NT; ++t) f[t] = wb_lds4(tile + t * WB_PLANE + off);
}
// Transposed operand: unit 32 m + (lane & 31), samples 16 t + 8 (lane >> 5) .. + 7, three planes (two transpose reads each)
PROMP_DEV void wb_read_tr(u32x4 (&f)[PROMP_NT], const float* tile, int lane, int m, int t) {
    const int p = lane & 15, ug = (lane >> 4) & 1, h = lane >> 5;
    const int c = 8 * m + 4 * ug + (p & 3);
    const int s0 = 16 * t + 8 * h + (p >> 2);
    const int o0 = wb_chunk(s0, c), o1 = wb_chunk(s0 + 4, c);
#pragma unroll
    for (int tt = 0; tt < PROMP_NT; ++tt) f[tt] = join_w2(lds_tr16(tile + tt * WB_PLANE + o0), lds_tr16(tile + tt * WB_PLANE + o1));
}
// The same for the 16x16x32 instruction: unit 16 ub16 + (lane & 15) (ub16 counts 16-unit blocks of the whole tile),
// samples 8 (lane >> 4) .. + 7
PROMP_DEV void wb_read_tr16(u32x4 (&f)[PROMP_NT], const float* tile, int lane, int ub16) {
    const int p = lane & 15, g = lane >> 4;
    const int c = 4 * ub16 + (p & 3);
    const int s0 = 8 * g + (p >> 2);
    const int o0 = wb_chunk(s0, c), o1 = wb_chunk(s0 + 4, c);
#pragma unroll
    for (int tt = 0; tt < PROMP_NT; ++tt) f[tt] = join_w2(lds_tr16(tile + tt * WB_PLANE + o0), lds_tr16(tile + tt * WB_PLANE + o1));
}
// The products of a split pair (promp_device.h: two FP16 terms: (1,0) (0,1) (0,0); three BF16 terms: the six with ta + tb <= 2),
// smallest first.  One accumulator chain: the matrix pipe forwards a result to the next instruction's C operand (same shape, same
// registers) without a bubble.  (The names keep the "6" of the BF16 form.)
#define WB_FOR_PRODUCTS(ta, tb) \
    _Pragma("unroll") for (int ta = PROMP_NT - 1; ta >= 0; --ta) _Pragma("unroll") for (int tb = PROMP_NT - 1 - ta; tb >= 0; --tb)
PROMP_DEV void wb_mma6(f32x16& c0, const u32x4 (&a)[PROMP_NT], const u32x4 (&b)[PROMP_NT]) {
    WB_FOR_PRODUCTS(ta, tb) c0 = mfma32_sw<PROMP_NT>(a[ta], b[tb], c0);
}
// two independent outputs against the same B operand, products interleaved (no instruction waits for its predecessor)
PROMP_DEV void wb_mma6_two(f32x16& ca, f32x16& cb, const u32x4 (&a)[PROMP_NT], const u32x4 (&b)[PROMP_NT], const u32x4 (&z)[PROMP_NT]) {
    WB_FOR_PRODUCTS(ta, tb) {
        ca = mfma32_sw<PROMP_NT>(a[ta], z[tb], ca);
        cb = mfma32_sw<PROMP_NT>(b[ta], z[tb], cb);
    }
}
// ca += a x za ;  cb += b x zb   (two unrelated products), interleaved
PROMP_DEV void wb_mma6_x2(f32x16& ca, const u32x4 (&a)[PROMP_NT], const u32x4 (&za)[PROMP_NT], f32x16& cb, const u32x4 (&b)[PROMP_NT],
                          const u32x4 (&zb)[PROMP_NT]) {
    WB_FOR_PRODUCTS(ta, tb) {
        ca = mfma32_sw<PROMP_NT>(a[ta], za[tb], ca);
        cb = mfma32_sw<PROMP_NT>(b[ta], zb[tb], cb);
    }
}
PROMP_DEV void wb_mma6_16(f32x4& c, const u32x4 (&a)[PROMP_NT], const u32x4 (&b)[PROMP_NT]) {
    WB_FOR_PRODUCTS(ta, tb) c = mfma16_sw<PROMP_NT>(a[ta], b[tb], c);
}
// the 16-bit pattern of 1.0 in both halves of a word (a row / column of ones rides in a gradient product)
PROMP_CX unsigned WB_ONE2 = PROMP_NT == 2 ? 0x3C003C00u : 0x3F803F80u;
PROMP_DEV f32x16 wb_zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
// out^T[own unit][sample] (+)= sum over NK steps of  W-planes (registers) x activation planes (LDS tile); the fragments of step
// q + 1 are requested before the products of step q
template <int NK>
PROMP_DEV void wb_chain(f32x16& c0, const u32x4 (&wpl)[NK][PROMP_NT], const float* tile, int j, int h) {
    u32x4 fb[PROMP_NT];
    wb_read_b(fb, tile, j, h, 0);
#pragma unroll
    for (int q = 0; q < NK; ++q) {
        u32x4 fn[PROMP_NT];
        if (q + 1 < NK) wb_read_b(fn, tile, j, h, q + 1);
        wb_mma6(c0, wpl[q], fb);
        if (q + 1 < NK) {
#pragma unroll
            for (int t = 0; t < PROMP_NT; ++t) fb[t] = fn[t];
        }
    }
}
// a D fragment (lane = sample j, units 32 w + 8 g + 4 h + i in register 4 g + i) -> the three planes of the wave's own columns
PROMP_DEV void wb_store_own(float* tile, const f32x16& v, int j, int h, int w) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned w0[PROMP_NT], w1[PROMP_NT];
        split_pair<PROMP_NT>(v[4 * g], v[4 * g + 1], w0);
        split_pair<PROMP_NT>(v[4 * g + 2], v[4 * g + 3], w1);
        const int off = wb_chunk(j, 8 * w + 2 * g + h);
#pragma unroll
        for (int t = 0; t < PROMP_NT; ++t) sts_w2(tile + t * WB_PLANE + off, w0[t], w1[t]);
    }
}
// one register quad of a D fragment (units 32 w + 8 g + 4 h + i of sample j) -> the three planes
PROMP_DEV void wb_store_quad(float* tile, float v0, float v1, float v2, float v3, int j, int h, int w, int g) {
    unsigned w0[PROMP_NT], w1[PROMP_NT];
    split_pair<PROMP_NT>(v0, v1, w0);
    split_pair<PROMP_NT>(v2, v3, w1);
    const int off = wb_chunk(j, 8 * w + 2 * g + h);
#pragma unroll
    for (int t = 0; t < PROMP_NT; ++t) sts_w2(tile + t * WB_PLANE + off, w0[t], w1[t]);
}
// a D fragment parked in / fetched from a wave-private, lane-linear LDS block (16-byte accesses, conflict-free)
PROMP_DEV void wb_park(float* blk, const f32x16& v, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 q;
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = v[4 * g + i];
        sts4(blk + 256 * g + 4 * lane, q);
    }
}
PROMP_DEV f32x16 wb_fetch(const float* blk, int lane) {
    f32x16 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 q = lds4(blk + 256 * g + 4 * lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[4 * g + i] = q[i];
    }
    return v;
}
// the bias of the wave's own units in D-fragment order
PROMP_DEV f32x16 wb_bias16(const float* bs, int h, int w) {
    f32x16 z;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = lds4(bs + 32 * w + 8 * g + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) z[4 * g + i] = b[i];
    }
    return z;
}
// a slice's planes of one K step from the pre-split copy (k_wb_planes): one 16-byte load per plane, lane-linear
PROMP_DEV void wb_gload(u32x4 (&pl)[PROMP_NT], const unsigned* P, int nk, int w, int q, int lane) {
#pragma unroll
    for (int t = 0; t < PROMP_NT; ++t) pl[t] = *(const u32x4*)(P + ((w * nk + q) * PROMP_NT + t) * 256 + 4 * lane);
}
// Output kernel, own 32 units: A operand of the 16x16x32 product  mu[a] = sum_u W3[32 w + u][a] h2[u]  (lane (a = l & 15, g): u = 8 g + e)
PROMP_DEV void wb_load_w3(u32x4 (&pl)[PROMP_NT], const float* W3, int A, int lane, int w, float sgn) {
    const int aa = lane & 15, g = lane >> 4;
    f32x4 lo, hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v0 = W3[(32 * w + 8 * g + e) * A + (aa < A ? aa : 0)], v1 = W3[(32 * w + 8 * g + 4 + e) * A + (aa < A ? aa : 0)];
        lo[e] = aa < A ? sgn * v0 : 0.f;
        hi[e] = aa < A ? sgn * v1 : 0.f;
    }
    pass_split8(lo, hi, pl);
}
// ... and as the A operand of the exact FP32 product  dH2[32 w + i] = sum_a W3[32 w + i][a] dmu[a]  (lane (i, h): a = 2 kk + h)
PROMP_DEV void wb_load_w3f(float (&r)[4], const float* W3, int A, int lane, int w, float sgn) {
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int aa = 2 * kk + h;
        const float v = W3[(32 * w + i) * A + (aa < A ? aa : 0)];
        r[kk] = aa < A ? sgn * v : 0.f;
    }
}
// One round's observations: requested into registers (wb_xreq: a round that does not exist reads a valid element and is all
// padding), split and stored as the three planes of an observation tile later (wb_xput: rows >= nrows and columns >= O zero,
// column OC one).  Thread t takes sample t / 8 and the chunks (units 4 c ..) c = t % 8 + 8 it: no division, and the eight threads
// of a sample read 128 contiguous bytes per step.
template <int NKO>
struct WbX {
    static constexpr int CPR = 4 * NKO, IT = (CPR + 7) / 8;
    float v[IT][4];
};
template <int NKO>
PROMP_DEV void wb_xreq(WbX<NKO>& X, const float* obs, long long row0, int nrows, int O, int tid) {
    // a wave-uniform 64-bit base and one 32-bit offset per element: rows past the round's last read its last row, columns past the
    // observation read its last column (valid addresses; wb_xput discards them)
    const float* rbase = obs + (nrows > 0 ? row0 * O : 0);
    const int last = nrows > 0 ? nrows - 1 : 0;
    const int s = (tid >> 3) < last ? (tid >> 3) : last;
    const unsigned ro = (unsigned)(s * O);
#pragma unroll
    for (int it = 0; it < WbX<NKO>::IT; ++it) {
        const int c = (tid & 7) + 8 * it;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = 4 * c + i;
            X.v[it][i] = rbase[ro + (unsigned)(o < O ? o : O - 1)];
        }
    }
}
template <int NKO>
PROMP_DEV void wb_xput_one(float* Xs, const WbX<NKO>& X, int it, int nrows, int O, int OC, int tid, float xs) {
    const int s = tid >> 3;
    const float rowm = s < nrows ? xs : 0.f;       // (xs: the power of two the task's observations are multiplied by, FP16 split)
    {
        const int c = (tid & 7) + 8 * it;
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = 4 * c + i;
            x[i] = (o == OC) ? 1.f : (o < O) ? rowm * X.v[it][i] : 0.f;
        }
        unsigned w0[PROMP_NT], w1[PROMP_NT];
        split_pair<PROMP_NT>(x[0], x[1], w0);
        split_pair<PROMP_NT>(x[2], x[3], w1);
        if (c < WbX<NKO>::CPR) {
            const int off = wb_chunk(s, c);
#pragma unroll
            for (int t = 0; t < PROMP_NT; ++t) sts_w2(Xs + t * WB_PLANE + off, w0[t], w1[t]);
        }
    }
}
template <int NKO>
PROMP_DEV void wb_xput(float* Xs, const WbX<NKO>& X, int nrows, int O, int OC, int tid, float xs) {
    const int s = tid >> 3;
    const float rowm = s < nrows ? xs : 0.f;
#pragma unroll
    for (int it = 0; it < WbX<NKO>::IT; ++it) {
        const int c = (tid & 7) + 8 * it;
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = 4 * c + i;
            x[i] = (o == OC) ? 1.f : (o < O) ? rowm * X.v[it][i] : 0.f;
        }
        unsigned w0[PROMP_NT], w1[PROMP_NT];
        split_pair<PROMP_NT>(x[0], x[1], w0);
        split_pair<PROMP_NT>(x[2], x[3], w1);
        if (c < WbX<NKO>::CPR) {
            const int off = wb_chunk(s, c);
#pragma unroll
            for (int t = 0; t < PROMP_NT; ++t) sts_w2(Xs + t * WB_PLANE + off, w0[t], w1[t]);
        }
    }
}
// zero planes, and the column of ones (unit OC) in plane 0
PROMP_DEV void wb_init_x(float* Xs, int OC, int tid) {
    for (int e = tid; e < WB_TILE; e += 256) ((unsigned*)Xs)[e] = 0u;
    __syncthreads();
    if (tid < WB_R) {
        const int off = wb_chunk(tid, OC >> 2) + ((OC & 3) >> 1);
        ((unsigned*)Xs)[off] = (OC & 1) ? (WB_ONE2 & 0xFFFF0000u) : (WB_ONE2 & 0xFFFFu);
    }
}

// K steps of a streamed slice: `G.r` holds the planes of WB_PF steps; wb_stream_begin requests steps 0 .. WB_PF - 1 (from anywhere
// ahead of the products), wb_chain_stream requests step q + WB_PF as soon as the products of step q have read their slot.
template <int PF>
struct WbRing {
    u32x4 r[PF][PROMP_NT];
};
// (FP16 split, k_wb_hvp walking a work item again: the direction's planes were made for a scale that turned out too large; `fix`, a
//  power of two below 1, brings a fragment down on its way into the registers.  1: the planes as they are.)
PROMP_DEV void wb_fix(u32x4 (&pl)[PROMP_NT], float fix) {
    if (PROMP_NT != 2 || fix == 1.f) return;
#pragma unroll
    for (int t = 0; t < PROMP_NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) pl[t][i] = pk_scale_f16(pl[t][i], fix);
}
template <int PF>
PROMP_DEV void wb_stream_begin(WbRing<PF>& G, const unsigned* P, int nk, int w, int lane, float fix = 1.f) {
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        wb_gload(G.r[q], P, nk, w, q, lane);
        wb_fix(G.r[q], fix);
    }
}
template <int NK, int WB_PF>
PROMP_DEV void wb_chain_stream(f32x16& c0, WbRing<WB_PF>& G, const unsigned* P, int w, int lane, const float* tile, int j, int h) {
    u32x4 fb[PROMP_NT];
    wb_read_b(fb, tile, j, h, 0);
#pragma unroll
    for (int q = 0; q < NK; ++q) {
        u32x4 fn[PROMP_NT];
        if (q + 1 < NK) wb_read_b(fn, tile, j, h, q + 1);
        sched_fence();
        wb_mma6(c0, G.r[q % WB_PF], fb);
        if (q + WB_PF < NK) wb_gload(G.r[q % WB_PF], P, NK, w, q + WB_PF, lane);
        sched_fence();
        if (q + 1 < NK) {
#pragma unroll
            for (int t = 0; t < PROMP_NT; ++t) fb[t] = fn[t];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_wb_planes: a parameter vector's two hidden kernels -> BF16 planes in A-fragment order (layout above).
// grid = (ceil(4 (NKO + 16) 64 / 256), tasks), block = 256: one thread per fragment (eight values, three planes).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_wb_planes(WbPlaneArgs a) {
    const int task = blockIdx.y;
    int r = blockIdx.x * 256 + threadIdx.x;
    const int n1 = 4 * a.NKO * 64, n2 = 4 * 8 * 64;
    if (r >= n1 + 2 * n2 + 256) return;
    const float* src = a.src + (long long)task * a.src_stride;
    unsigned* dst = a.dst + (long long)task * wb_planes_words(a.NKO);
    const int oW2a = a.O * 128 + 128;
    int region = 0;
    const float vsc = wb_vec_scale(a.vec_absmax, task, WB_V_TARGET), w1sc = pow2f(wb_obs_shift(a.obs_absmax, task)) * vsc;
    if (r >= n1 + 2 * n2) {               // the output kernel's fragments (wb_load_w3)
        r -= n1 + 2 * n2;
        u32x4 pl[PROMP_NT];
        wb_load_w3(pl, src + oW2a + 128 * 128 + 128, a.A, r & 63, r >> 6, vsc);
#pragma unroll
        for (int t = 0; t < PROMP_NT; ++t) *(u32x4*)(dst + wb_planes_w3(a.NKO) + ((r >> 6) * PROMP_NT + t) * 256 + 4 * (r & 63)) = pl[t];
        return;
    }
    if (r >= n1 + n2) { region = 2; r -= n1 + n2; dst += wb_planes_r2(a.NKO); }
    else if (r >= n1) { region = 1; r -= n1; dst += wb_planes_c2(a.NKO); }
    const int NK = region == 0 ? a.NKO : 8;
    const int lane = r & 63, wq = r >> 6, w = wq / NK, q = wq - w * NK;
    const int i = lane & 31, h = lane >> 5;
    const int oW2 = a.O * 128 + 128;
    f32x4 lo, hi;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 16 * q + 8 * h + e;
        float x;
        if (region == 0) x = k < a.O ? w1sc * src[k * 128 + 32 * w + i] : 0.f;
        else if (region == 1) x = vsc * src[oW2 + k * 128 + 32 * w + i];
        else x = (a.row_sign * vsc) * src[oW2 + (32 * w + i) * 128 + k];
        if (e < 4) lo[e] = x;
        else hi[e - 4] = x;
    }
    u32x4 pl[PROMP_NT];
    pass_split8(lo, hi, pl);
#pragma unroll
    for (int t = 0; t < PROMP_NT; ++t) *(u32x4*)(dst + ((w * NK + q) * PROMP_NT + t) * 256 + 4 * lane) = pl[t];
}

// The largest |entry| of every task's vector (the direction of an R-operator pass) -> out[task], bits of a non-negative float
// (k_wb_planes and k_wb_hvp take the direction's scale from it: wb_vec_scale).  grid = tasks, block = 256.
struct VecAbsmaxArgs {
    const float* src;
    long long stride;
    int n, n_w1;                 // entries; the leading ones that belong to the hidden_0 kernel
    const float* obs_absmax;     // [tasks] or NULL: the hidden_0 block is weighed by the inverse of the observations' scale
    unsigned* out;               // (chain_stage_nets: the blocks are compared by what they contribute to a tangent)
};
__global__ void __launch_bounds__(1024) k_vec_absmax(VecAbsmaxArgs a) {
    const int task = blockIdx.x, tid = threadIdx.x;
    const float* src = a.src + (long long)task * a.stride;
    const float w1w = pow2f(obs_shift(a.obs_absmax, task));
    // four loads in flight per thread and trip (one workgroup of 16 waves per task: the vector is ~30 k floats)
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    int i = tid;
    for (; i + 3072 < a.n; i += 4096) {
        const float x0 = src[i], x1 = src[i + 1024], x2 = src[i + 2048], x3 = src[i + 3072];
        m0 = fmaxf(m0, (i < a.n_w1 ? w1w : 1.f) * fabsf(x0));
        m1 = fmaxf(m1, (i + 1024 < a.n_w1 ? w1w : 1.f) * fabsf(x1));
        m2 = fmaxf(m2, (i + 2048 < a.n_w1 ? w1w : 1.f) * fabsf(x2));
        m3 = fmaxf(m3, (i + 3072 < a.n_w1 ? w1w : 1.f) * fabsf(x3));
    }
    for (; i < a.n; i += 1024) m0 = fmaxf(m0, (i < a.n_w1 ? w1w : 1.f) * fabsf(src[i]));
    const float m = wave_absmax_f32(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)));
    PROMP_SMEM_DECL;
    float* sm = (float*)PROMP_SMEM_PTR;
    if ((tid & 63) == 0) sm[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t = fmaxf(t, sm[w]);
        a.out[task] = __builtin_bit_cast(unsigned, t);
    }
}

// ---------------------------------------------------------------------------------------------
// k_wb_fwd_bwd: objective (+ gradient) of one policy pass; PassArgs / partial layout of k_wide_fwd_bwd.
// grid = work items (table 0), block = 256 (4 waves, one per SIMD), one workgroup per CU.
// NKO = K = 16 steps of the observation (obs_dim <= 16 NKO), NXB = 32-unit blocks of the observation incl. the column of ones at
// unit OC (the first free unit of the last K step, or of the last block: obs_dim <= OC).
// ---------------------------------------------------------------------------------------------
template <int NKO, int NXB, bool BWD>
__global__ void __launch_bounds__(256, 1) k_wb_fwd_bwd(PassArgs a) {
    constexpr int H = 128, R = WB_R, MS = WB_MS, OC = (32 * NXB > 16 * NKO ? 32 * NXB : 16 * NKO) - 1;
    PROMP_SMEM_DECL;
    float* sm = (float*)PROMP_SMEM_PTR;
    const int w = wave_uniform(threadIdx.x >> 6);
    const int wi = xcd_item(blockIdx.x, gridDim.x);
    const WorkItem wk = a.work[wi];
    const int task = wk.task;
    const int O = a.O, A = a.A;
    const int ob1 = O * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW3 = ob2 + H, ob3 = oW3 + H * A, oS = ob3 + A, NP = oS + A;
    const LdsWB L = wb_layout(false);
    float *Mp = sm + L.mp, *lmask = sm + L.lmask, *red = sm + L.red;
    const int ntask = a.task_row_offsets[task + 1] - a.task_row_offsets[task];
    const float invN = 1.0f / (float)ntask;
    const float* th = a.theta + (long long)task * a.theta_task_stride;
    const unsigned* PL = a.wb_theta_planes + (long long)task * a.wb_plane_stride;
    const unsigned* PR2 = PL + wb_planes_r2(NKO);
    // FP16 split: the task's observations times 2^-sx (k_wb_planes multiplied the hidden_0 kernel by 2^sx)
    const int sx = wb_obs_shift(a.obs_absmax, task);
    const float xs = pow2f(-sx), w1u = pow2f(sx);

    // ---- forward weight planes of this lane, resident for the whole work item; the first round's observations
    u32x4 w1p[NKO][PROMP_NT], w2c[8][PROMP_NT];
    float w3f[4];
    WbX<NKO> X;
    {
        const int tid = threadIdx.x, lane = tid & 63;
        float *Xs = sm + L.x0, *Mss = sm + L.ms, *Dm = sm + L.dm, *b1s = sm + L.b1, *b2s = sm + L.b2, *b3s = sm + L.b3, *lss = sm + L.ls,
              *ess = sm + L.es, *sn2s = sm + L.sn2;
        wb_xreq<NKO>(X, a.obs, wk.row_begin, wk.row_end - wk.row_begin, O, tid);
#pragma unroll
        for (int q = 0; q < NKO; ++q) wb_gload(w1p[q], PL, NKO, w, q, lane);
#pragma unroll
        for (int q = 0; q < 8; ++q) wb_gload(w2c[q], PL + wb_planes_c2(NKO), 8, w, q, lane);
        if (BWD) wb_load_w3f(w3f, th + oW3, A, lane, w, 1.f);
        for (int e = tid; e < H; e += 256) {
            b1s[e] = th[ob1 + e];
            b2s[e] = th[ob2 + e];
        }
        if (tid < 16) {
            const float sr = (tid < A) ? th[oS + tid] : 0.f;
            const bool clipped = a.clip_log_std && (sr < a.min_log_std);   // tf.maximum: gradient iff var >= min
            const float s = clipped ? a.min_log_std : sr;
            lss[tid] = s;
            lmask[tid] = clipped ? 0.f : 1.f;
            ess[tid] = expf(-s);
            sn2s[tid] = expf(2.f * s);
            b3s[tid] = (tid < A) ? th[ob3 + tid] : 0.f;
        }
        for (int e = tid; e < R * MS; e += 256) Mss[e] = 0.f;
        for (int e = tid; e < PROMP_NT * WB_DMPL; e += 256) ((unsigned*)Dm)[e] = 0u;
        wb_init_x(Xs, OC, tid);
        wb_init_x(sm + L.x1, OC, tid);
        wb_xput<NKO>(Xs, X, wk.row_end - wk.row_begin, O, OC, tid, xs);
        wb_xreq<NKO>(X, a.obs, wk.row_begin + R, wk.row_end - wk.row_begin - R, O, tid);
    }

    // gradient slices: hidden_0 kernel [32 m + ..][32 w + j] (row OC: the hidden_0 bias gradient), hidden_1 kernel [32 m + ..][32 w + j],
    // output kernel rows 32 w + 16 ub + (lane & 15) (row 15 of its action axis: the hidden_1 bias gradient)
    f32x16 aw1[NXB], aw2[4];
    f32x4 aw3[2];
    float loss, klsum, gs, gb3;
    // FP16 split (promp_kernels_pass.h: pass_cotangent_scale; here the four waves share every tile, so the scale is the
    // workgroup's): the cotangents carry cs, set by the first round that has one; the work item is walked again, with its largest
    // cotangent for a scale, if a split left the format (`attempt`)
    float cs = 1.f, ics = 1.f, amax = 0.f, redo_amax = 0.f;
    int prov = 1;
    for (int attempt = 0;; ++attempt) {
    if (attempt > 0) {           // the first two rounds' observations again
        const int tid = threadIdx.x;
        __syncthreads();
        wb_xreq<NKO>(X, a.obs, wk.row_begin, wk.row_end - wk.row_begin, O, tid);
        wb_xput<NKO>(sm + L.x0, X, wk.row_end - wk.row_begin, O, OC, tid, xs);
        wb_xreq<NKO>(X, a.obs, wk.row_begin + R, wk.row_end - wk.row_begin - R, O, tid);
    }
#pragma unroll
    for (int m = 0; m < NXB; ++m) aw1[m] = wb_zero16();
#pragma unroll
    for (int m = 0; m < 4; ++m) aw2[m] = wb_zero16();
    aw3[0] = aw3[1] = zero4();
    loss = klsum = gs = gb3 = 0.f;
    cs = ics = 1.f;
    amax = 0.f;
    prov = 1;
    if (PROMP_NT == 2 && attempt > 0 && redo_amax > 0.f && redo_amax < 3.0e38f) {
        int k = scale_exp(redo_amax, PASS_CT_REDO - (attempt - 1) * PASS_CT_RETRY);
        k = k < -100 ? -100 : k > 100 ? 100 : k;
        cs = pow2f(k);
        ics = pow2f(-k);
        prov = 0;
    }

    int rix = 0;
    for (int base = wk.row_begin; base < wk.row_end; base += R) {
        const int nrows = (wk.row_end - base) < R ? (wk.row_end - base) : R;
        // loop-variant lane indices and LDS base (opaque_zero): the lane-constant address arithmetic of the round is recomputed where
        // it is used instead of being hoisted out of the loop and kept (spilled) across it
        const int zr = opaque_zero();
        const int tid = threadIdx.x + zr, lane = tid & 63, j = lane & 31, h = lane >> 5;
        float* const smz = sm + zr;
        float *Xs = smz + ((rix & 1) ? L.x1 : L.x0), *Xn = smz + ((rix & 1) ? L.x0 : L.x1), *H1s = smz + L.h1, *H2s = smz + L.h2,
              *DZ1 = smz + L.dz1, *Mp = smz + L.mp, *Mss = smz + L.ms, *Dm = smz + L.dm, *b1s = smz + L.b1, *b2s = smz + L.b2,
              *b3s = smz + L.b3, *lss = smz + L.ls, *ess = smz + L.es, *sn2s = smz + L.sn2, *D1p = smz + L.d1 + w * 1024;
        WB_STAMP(0);
        __syncthreads();                       // the previous round is done with X / H1 / H2 / the mean tiles; this round's X is complete
        WB_STAMP(1);
        // ---- the distribution epilogue's row data, requested now: one (row, action) pair per thread
        const int erow = tid >> 3, eq = tid & 7;
        const bool eown = eq < A, rvalid = erow < nrows;
        float advn, eac, emo, eso;
        {
            const long long n = (long long)base + (rvalid ? erow : 0);
            const int qq = eown ? eq : 0;
            const float* olsp = a.old_log_std + (a.ls_per_row ? n * A : (long long)task * A);
            advn = a.adv[n];
            eac = a.act[n * A + qq];
            emo = a.old_mean[n * A + qq];
            eso = olsp[qq];
        }
        // ---- layer 1: H1 = tanh(W1^T X^T + b1), own units.  The NEXT round's observations (requested a round ago, in registers) are
        //      split into the other observation tile chunk by chunk among the chain's products: vector work in the matrix
        //      instructions' shadow (it used to run behind the layer: 2.2 k of a round's 21.7 k cycles)
        {
            f32x16 c0 = wb_bias16(b1s, h, w);
            u32x4 fb[PROMP_NT];
            wb_read_b(fb, Xs, j, h, 0);
#pragma unroll
            for (int q = 0; q < NKO; ++q) {
                u32x4 fn[PROMP_NT];
                if (q + 1 < NKO) wb_read_b(fn, Xs, j, h, q + 1);
                if (q < WbX<NKO>::IT) wb_xput_one<NKO>(Xn, X, q, wk.row_end - base - R, O, OC, tid, xs);
                wb_mma6(c0, w1p[q], fb);
                if (q + 1 < NKO) {
#pragma unroll
                    for (int t = 0; t < PROMP_NT; ++t) fb[t] = fn[t];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) c0[r] = fast_tanh(c0[r]);
            wb_store_own(H1s, c0, j, h, w);
            if (BWD) {
#pragma unroll
                for (int r = 0; r < 16; ++r) c0[r] = 1.f - c0[r] * c0[r];
                wb_park(D1p, c0, lane);
            }
        }
        sched_fence();
        WB_STAMP(2);
        WB_STAMP(21);
        // ---- the round after the next is requested
        wb_xreq<NKO>(X, a.obs, (long long)base + 2 * R, wk.row_end - base - 2 * R, O, tid);
        WB_STAMP(3);
        __syncthreads();
        WB_STAMP(4);
        // ---- layer 2
        f32x16 d2;
        {
            f32x16 c0 = wb_bias16(b2s, h, w);
            wb_chain<8>(c0, w2c, H1s, j, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float hv = fast_tanh(c0[r]);
                c0[r] = hv;
                d2[r] = 1.f - hv * hv;
            }
            wb_store_own(H2s, c0, j, h, w);
        }
        wave_sync();
        sched_fence();
        WB_STAMP(5);
        // ---- output layer: this wave's 32 units of the contraction, both 16-sample blocks -> partial means
        {
            const int j16 = lane & 15, g4 = lane >> 4;
            u32x4 w3p[PROMP_NT];
            wb_gload(w3p, PL + zr + wb_planes_w3(NKO), 1, w, 0, lane);
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const int s = 16 * sb + j16;
                u32x4 fb[PROMP_NT];
                const int off = wb_chunk(s, 8 * w + 2 * g4);
#pragma unroll
                for (int t = 0; t < PROMP_NT; ++t) fb[t] = wb_lds4(H2s + t * WB_PLANE + off);
                f32x4 dm = zero4();
                wb_mma6_16(dm, w3p, fb);
                if (g4 < 2) sts4(Mp + (w * R + s) * 8 + 4 * g4, dm);      // rows 4 g4 + r = actions
            }
        }
        // the backward product's row planes: the first steps are requested here, a barrier and an epilogue ahead of their products
        WbRing<3> G;
        if (BWD) wb_stream_begin(G, PR2, 8, w, lane);
        WB_STAMP(6);
        __syncthreads();
        WB_STAMP(7);
        // ---- distribution + objective epilogue (the arithmetic of k_wide_fwd_bwd, one (row, action) pair per thread: the sums over a
        //      row's actions run over its 8 lanes)
        {
            float dlp = 0.f, sumz2 = 0.f, sums = 0.f, kl = 0.f, z = 0.f, ee = 0.f, dklm = 0.f, dkls = 0.f;
            if (eown) {
                const float mu = b3s[eq] + ((Mp[(0 * R + erow) * 8 + eq] + Mp[(1 * R + erow) * 8 + eq]) + (Mp[(2 * R + erow) * 8 + eq] + Mp[(3 * R + erow) * 8 + eq]));
                const float s = lss[eq];
                ee = ess[eq];
                z = (eac - mu) * ee;
                const float zo = (eac - emo) * fast_exp(-eso);
                dlp = (eso - s) - 0.5f * (z * z - zo * zo);
                sumz2 = z * z;
                sums = s;
                const float sn2 = sn2s[eq], num = (emo - mu) * (emo - mu) + fast_exp(2.f * eso) - sn2, den = 2.f * sn2 + 1e-8f;
                const float rden = fast_rcp(den);
                kl = num * rden + s - eso;
                dklm = -2.f * (emo - mu) * rden;
                dkls = (-2.f * sn2 * den - 4.f * num * sn2) * (rden * rden) + 1.f;
            }
#pragma unroll
            for (int m = 1; m <= 4; m <<= 1) {
                dlp += shfl_xor_f32(dlp, m);
                sumz2 += shfl_xor_f32(sumz2, m);
                sums += shfl_xor_f32(sums, m);
                kl += shfl_xor_f32(kl, m);
            }
            float c = 0.f, ck = 0.f;
            if (rvalid) {
                const float rho = expf(dlp);
                float lrow;
                if (a.loss_kind == LOSS_KL) {
                    lrow = kl * invN;
                    ck = invN;
                } else if (a.loss_kind == LOSS_RATIO) {
                    lrow = -rho * advn * invN;
                    c = -advn * rho * invN;
                } else if (a.loss_kind == LOSS_CLIP) {
                    const float x = rho * advn;
                    const float y = fminf(fmaxf(rho, 1.f - a.clip_eps), 1.f + a.clip_eps) * advn;
                    lrow = -fminf(x, y) * invN;
                    c = (x <= y) ? -advn * rho * invN : 0.f;
                } else {
                    const float lp = -sums - 0.5f * sumz2 - 0.5f * (float)A * 1.8378770664093453f;
                    lrow = -lp * advn * invN;
                    c = -advn * invN;
                }
                if (eq == 0) {
                    loss += lrow;
                    klsum += kl * invN;
                }
            }
            float d = 0.f;
            if (BWD && eown) {
                d = c * z * ee + ck * dklm;
                gs += c * (z * z - 1.f) + ck * dkls;
                gb3 += d;
            }
            if (BWD && PROMP_NT == 2) {
                amax = fmaxf(amax, fabsf(d));
                if (prov) {          // (the same in every thread) no round has had a cotangent yet: this one sets the scale
                    const float m = wave_absmax_f32(d);
                    if (lane == 0) red[w] = m;
                    __syncthreads();
                    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
                    const bool okm = mx > 0.f && mx < 3.0e38f;
                    int k = scale_exp(okm ? mx : invN, okm ? PASS_CT_TARGET : -4);
                    k = k < -100 ? -100 : k > 100 ? 100 : k;
                    cs = pow2f(k);
                    ics = pow2f(-k);
                    prov = okm ? 0 : 1;
                }
                d *= cs;
            }
            if (BWD && eown) {
                // the cotangent of the mean: float32 for the exact W3 product, the split's planes for the output-kernel gradient
                Mss[erow * MS + eq] = d;
                unsigned t3[PROMP_NT];
                split_pair<PROMP_NT>(d, 0.f, t3);
                unsigned short* Dh = (unsigned short*)Dm;
#pragma unroll
                for (int t = 0; t < PROMP_NT; ++t) Dh[2 * (t * WB_DMPL + erow * WB_DMROW) + eq] = (unsigned short)(t3[t] & 0xFFFFu);
            }
        }
        WB_STAMP(8);
        if (!BWD) {
            ++rix;
            continue;
        }
        __syncthreads();
        WB_STAMP(9);
        // ---- output-kernel gradient rows 32 w .. (+=), the hidden_1 bias gradient in its row 15; dZ2 = (W3 dmu^T) (1 - H2^2)
        {
            const int p = lane & 15, g4 = lane >> 4;
            // A = dmu^T: action l & 15, samples 8 g4 .. + 7
            u32x4 fa[PROMP_NT];
            {
                const int o0 = (8 * g4 + (p >> 2)) * WB_DMROW + 2 * (p & 3), o1 = o0 + 4 * WB_DMROW;
#pragma unroll
                for (int t = 0; t < PROMP_NT; ++t) fa[t] = join_w2(lds_tr16(Dm + t * WB_DMPL + o0), lds_tr16(Dm + t * WB_DMPL + o1));
            }
#pragma unroll
            for (int ub = 0; ub < 2; ++ub) {
                u32x4 fb[PROMP_NT];
                wb_read_tr16(fb, H2s, lane, 2 * w + ub);
                wb_mma6_16(aw3[ub], fa, fb);
            }
            f32x16 dh2 = wb_zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dh2 = mfma32(w3f[kk], Mss[j * MS + 2 * kk + h], dh2);
#pragma unroll
            for (int r = 0; r < 16; ++r) dh2[r] *= d2[r];
            wave_sync();                       // this wave's own reads of its H2 columns (above) precede the overwrite
            wb_store_own(H2s, dh2, j, h, w);
            wave_sync();
            // hidden_1 bias gradient: a row of ones (action slot 15) against the wave's own dZ2 columns
            u32x4 ones;
#pragma unroll
            for (int i = 0; i < 4; ++i) ones[i] = (p == 15) ? WB_ONE2 : 0u;
#pragma unroll
            for (int ub = 0; ub < 2; ++ub) {
                u32x4 fb[PROMP_NT];
                wb_read_tr16(fb, H2s, lane, 2 * w + ub);
#pragma unroll
                for (int t = PROMP_NT - 1; t >= 0; --t) aw3[ub] = mfma16_sw<PROMP_NT>(ones, fb[t], aw3[ub]);
            }
        }
        sched_fence();
        WB_STAMP(10);
        __syncthreads();
        WB_STAMP(11);
        // ---- dH1 = W2 dZ2^T for the own units (row planes streamed) with the hidden_1 kernel gradient columns 32 w .. (+=) in its
        //      shadow: one block and sample half per K step (a K step is then 12 products long, and the ring's 3 steps of lead
        //      cover an L2 round trip); dZ1 -> its own tile
        {
            f32x16 c0 = wb_zero16();
            u32x4 fb[PROMP_NT], fz[PROMP_NT];
            wb_read_b(fb, H2s, j, h, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                u32x4 fn[PROMP_NT], fa[PROMP_NT];
                if (q + 1 < 8) wb_read_b(fn, H2s, j, h, q + 1);
                const int t = q >> 2, m = q & 3;
                if (m == 0) wb_read_tr(fz, H2s, lane, w, t);
                wb_read_tr(fa, H1s, lane, m, t);
                sched_fence();        // the ring's loads must not sink to their uses: a K step's requests, then its dense product block
                wb_mma6_x2(c0, G.r[q % 3], fb, aw2[m], fa, fz);
                if (q + 3 < 8) wb_gload(G.r[q % 3], PR2, 8, w, q + 3, lane);
                sched_fence();
                if (q + 1 < 8) {
#pragma unroll
                    for (int tt = 0; tt < PROMP_NT; ++tt) fb[tt] = fn[tt];
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 d1 = lds4(D1p + 256 * g + 4 * lane);
#pragma unroll
                for (int i = 0; i < 4; ++i) c0[4 * g + i] *= d1[i];
            }
            wb_store_own(DZ1, c0, j, h, w);
        }
        wave_sync();
        sched_fence();
        WB_STAMP(12);
        WB_STAMP(13);
        // ---- hidden_0 kernel gradient columns 32 w .. (+=): needs only this wave's own dZ1 columns
        {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4 fz[PROMP_NT];
                wb_read_tr(fz, DZ1, lane, w, t);
                if (NXB == 1) {
                    u32x4 fa[PROMP_NT];
                    wb_read_tr(fa, Xs, lane, 0, t);
                    wb_mma6(aw1[0], fa, fz);
                } else {
#pragma unroll
                    for (int m = 0; m + 1 < NXB; m += 2) {
                        u32x4 fa0[PROMP_NT], fa1[PROMP_NT];
                        wb_read_tr(fa0, Xs, lane, m, t);
                        wb_read_tr(fa1, Xs, lane, m + 1, t);
                        wb_mma6_two(aw1[m], aw1[m + 1], fa0, fa1, fz);
                    }
                }
            }
        }
        WB_STAMP(14);
        ++rix;
    }
    // FP16 split: did every split stay inside the format?  The mean cotangents' planes (they feed the output kernel's sums only): their
    // largest value at the scale; everything else ends in the hidden_0 kernel sums (x * 0 is 0 for finite x only).
    if (!BWD || PROMP_NT != 2 || attempt + 1 >= PASS_CT_ATTEMPTS) break;
    {
        float chk = 0.f;
#pragma unroll
        for (int m = 0; m < NXB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) chk = __builtin_fmaf(aw1[m][r], 0.f, chk);
        const float wm = wave_absmax_f32(amax), wbad = wave_any(chk != chk) ? 1.f : 0.f;
        const int lane = threadIdx.x & 63;
        __syncthreads();
        if (lane == 0) {
            red[w] = wm;
            red[4 + w] = wbad;
        }
        __syncthreads();
        redo_amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const bool bad = (red[4] + red[5] + red[6] + red[7]) > 0.f || !(redo_amax * cs <= 65504.f);
        if (!bad) break;
        if (threadIdx.x == 0) atomic_add_agent(a.split_events + 0, 1);
    }
    }

    // ---- results: scalars through LDS in wave order, gradient slices straight from their owners ----
    float* P = a.partials + (long long)wi * a.partial_stride;
    const int tidt = threadIdx.x + opaque_zero(), lane = tidt & 63, j = lane & 31, h = lane >> 5, tid = tidt;
    // per-action sums of the epilogue lanes (action = lane & 7): over the wave's 8 rows, then over the waves in wave order
#pragma unroll
    for (int m = 8; m <= 32; m <<= 1) {
        gs += shfl_xor_f32(gs, m);
        gb3 += shfl_xor_f32(gb3, m);
        loss += shfl_xor_f32(loss, m);
        klsum += shfl_xor_f32(klsum, m);
    }
    __syncthreads();
    if (lane < 8) {
        float* rw = red + 16 * w;
        rw[lane] = gs;
        rw[8 + lane] = gb3;
    }
    float* sc = Mp;                // two scalars per wave
    if (lane == 0) {
        sc[2 * w] = loss;
        sc[2 * w + 1] = klsum;
    }
    __syncthreads();
    if (tid == 0) {
        P[NP] = (sc[0] + sc[2]) + (sc[4] + sc[6]);
        P[NP + 1] = (sc[1] + sc[3]) + (sc[5] + sc[7]);
    }
    if (!BWD) return;
    if (tid < 16) {
        const int aidx = tid & 7, which = tid >> 3;     // which: 0 log-std gradient, 1 output bias gradient
        const float t = (red[8 * which + aidx] + red[16 + 8 * which + aidx]) + (red[32 + 8 * which + aidx] + red[48 + 8 * which + aidx]);
        if (aidx < A) {
            if (which == 0) P[oS + aidx] = t * lmask[aidx];
            else P[ob3 + aidx] = t;
        }
    }
    // hidden_1 kernel gradient: block m, register 4 g + i <-> row 32 m + 8 g + 4 h + i, column 32 w + j
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) P[oW2 + (32 * m + 8 * (r >> 2) + 4 * h + (r & 3)) * H + 32 * w + j] = aw2[m][r] * ics;
    // hidden_0 kernel gradient; row OC = the hidden_0 bias gradient
#pragma unroll
    for (int m = 0; m < NXB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * m + 8 * (r >> 2) + 4 * h + (r & 3);
            if (row < O) P[row * H + 32 * w + j] = aw1[m][r] * (ics * w1u);     // (the observations' scale comes off here)
            else if (row == OC) P[ob1 + 32 * w + j] = aw1[m][r] * ics;
        }
    // output kernel rows 32 w + 16 ub + (lane & 15): D rows 4 g4 + r = actions; action slot 15 = hidden_1 bias gradient
    {
        const int j16 = lane & 15, g4 = lane >> 4;
#pragma unroll
        for (int ub = 0; ub < 2; ++ub)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int aa = 4 * g4 + r, unit = 32 * w + 16 * ub + j16;
                if (aa < A) P[oW3 + unit * A + aa] = aw3[ub][r] * ics;
                else if (aa == 15) P[ob2 + unit] = aw3[ub][r] * ics;
            }
    }
}

// ca += a x ba ;  cb += a x bb   (one weight slice against two activation tensors), products interleaved
PROMP_DEV void wb_mma6_ab(f32x16& ca, f32x16& cb, const u32x4 (&a)[PROMP_NT], const u32x4 (&ba)[PROMP_NT], const u32x4 (&bb)[PROMP_NT]) {
    WB_FOR_PRODUCTS(ta, tb) {
        ca = mfma32_sw<PROMP_NT>(a[ta], ba[tb], ca);
        cb = mfma32_sw<PROMP_NT>(a[ta], bb[tb], cb);
    }
}
// The ring runs WB_PF (its depth) steps ahead ACROSS slices: behind the products of step q of a slice of NK steps, slot q % WB_PF takes step
// q + WB_PF of the same slice or, in the slice's last WB_PF steps, step q % WB_PF of the NEXT slice the wave will walk (which then
// finds its first WB_PF steps in slots 0 .. WB_PF - 1).  Needs WB_PF <= the steps of every slice.
template <int WB_PF>
PROMP_DEV void wb_ring_next(WbRing<WB_PF>& G, const unsigned* Pcur, int nk, const unsigned* Pnext, int nkn, int w, int lane, int q, float fix = 1.f) {
    if (q + WB_PF < nk) wb_gload(G.r[q % WB_PF], Pcur, nk, w, q + WB_PF, lane);
    else wb_gload(G.r[q % WB_PF], Pnext, nkn, w, q % WB_PF, lane);
    wb_fix(G.r[q % WB_PF], fix);
}

// ---------------------------------------------------------------------------------------------
// k_wb_hvp:  out = -(d^2 L/d theta^2) v + kl_weight * grad KL  (the quantities of k_wide_hvp, its PassArgs and partial layout),
// or with LOSS_KL  -(d^2 KL/d theta^2) v  (TRPO's exact constraint product).
// Every weight slice is streamed from the pre-split copies of theta and of the direction (k_wb_planes): hidden_0 columns,
// hidden_1 columns, hidden_1 rows, each for both vectors, through two register rings that run WB_PF K steps ahead across the phases.
// The cotangents are kept NEGATED (nd = -dmu, ndZ2 = -dZ2) so that every product with the direction's planes is an addition.
// grid = work items (table 0), block = 256.
// ---------------------------------------------------------------------------------------------
template <int NKO, int NXB>
__global__ void __launch_bounds__(256, 1) k_wb_hvp(PassArgs a) {
    constexpr int H = 128, R = WB_R, MS = WB_MS, OC = (32 * NXB > 16 * NKO ? 32 * NXB : 16 * NKO) - 1, WB_PF = 3;
    static_assert(NKO >= WB_PF, "the weight ring needs WB_PF <= K steps of every slice");
    PROMP_SMEM_DECL;
    float* sm = (float*)PROMP_SMEM_PTR;
    const int w = wave_uniform(threadIdx.x >> 6);
    const int wi = xcd_item(blockIdx.x, gridDim.x);
    const WorkItem wk = a.work[wi];
    const int task = wk.task;
    const int O = a.O, A = a.A;
    const int ob1 = O * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW3 = ob2 + H, ob3 = oW3 + H * A, oS = ob3 + A, NP = oS + A;
    const LdsWB L = wb_layout(true);
    float *Mp = sm + L.mp, *lmask = sm + L.lmask, *red = sm + L.red;
    const int ntask = a.task_row_offsets[task + 1] - a.task_row_offsets[task];
    const float invN = 1.0f / (float)ntask;
    const float* th = a.theta + (long long)task * a.theta_task_stride;
    const float* vv = a.vdir + (long long)task * NP;
    const unsigned* PT = a.wb_theta_planes + (long long)task * a.wb_plane_stride;
    const unsigned* PV = a.wb_v_planes + (long long)task * wb_planes_words(NKO);
    const int oC2 = wb_planes_c2(NKO), oR2 = wb_planes_r2(NKO), oW3P = wb_planes_w3(NKO);
    // FP16 split: the task's observations times 2^-sx (k_wb_planes multiplied both hidden_0 kernels by 2^sx); the direction times vs
    // (k_wb_planes did its kernels; the biases, the output kernel's float32 copy and log_std follow here) -- every tangent carries vs
    const int sx = wb_obs_shift(a.obs_absmax, task);
    const float xs = pow2f(-sx), w1u = pow2f(sx);
    const float vs0 = wb_vec_scale(a.vdir_absmax, task, WB_V_TARGET);
    float vs = vs0, ivs = 1.f / vs0, vfix = 1.f;      // (vfix: see `attempt`)

    float w3f[4], v3f[4];
    WbX<NKO> X;
    WbRing<WB_PF> GT, GV;
    {
        const int tid = threadIdx.x, lane = tid & 63;
        float *Xs = sm + L.x0, *Mss = sm + L.ms, *Ms2s = sm + L.ms2, *Dm = sm + L.dm, *Dm2 = sm + L.dm2, *b1s = sm + L.b1, *b2s = sm + L.b2,
              *b3s = sm + L.b3, *vb1s = sm + L.vb1, *vb2s = sm + L.vb2, *vb3s = sm + L.vb3, *lss = sm + L.ls, *ess = sm + L.es,
              *sn2s = sm + L.sn2, *vls = sm + L.vls;
        wb_xreq<NKO>(X, a.obs, wk.row_begin, wk.row_end - wk.row_begin, O, tid);
        wb_stream_begin(GT, PT, NKO, w, lane);
        wb_stream_begin(GV, PV, NKO, w, lane);
        wb_load_w3f(w3f, th + oW3, A, lane, w, 1.f);
        wb_load_w3f(v3f, vv + oW3, A, lane, w, vs);
        for (int e = tid; e < H; e += 256) {
            b1s[e] = th[ob1 + e];
            b2s[e] = th[ob2 + e];
            vb1s[e] = vs * vv[ob1 + e];
            vb2s[e] = vs * vv[ob2 + e];
        }
        if (tid < 16) {
            const float sr = (tid < A) ? th[oS + tid] : 0.f;
            const bool clipped = a.clip_log_std && (sr < a.min_log_std);
            const float s = clipped ? a.min_log_std : sr;
            lss[tid] = s;
            lmask[tid] = clipped ? 0.f : 1.f;
            ess[tid] = expf(-s);
            sn2s[tid] = expf(2.f * s);
            vls[tid] = (tid < A && !clipped) ? vs * vv[oS + tid] : 0.f;   // R{s} = mask * v_s
            b3s[tid] = (tid < A) ? th[ob3 + tid] : 0.f;
            vb3s[tid] = (tid < A) ? vs * vv[ob3 + tid] : 0.f;
        }
        for (int e = tid; e < R * MS; e += 256) Mss[e] = Ms2s[e] = 0.f;
        for (int e = tid; e < PROMP_NT * WB_DMPL; e += 256) ((unsigned*)Dm)[e] = ((unsigned*)Dm2)[e] = 0u;
        wb_init_x(Xs, OC, tid);
    }

    f32x16 aw1[NXB], aw2[4];
    f32x4 aw3[2];
    float klsum, outs, outb3;
    float klv = a.kl_weight * vs;            // (the KL cotangents carry no factor of the direction: they take its scale here)
    // FP16 split (k_wb_fwd_bwd; promp_kernels_chain.h: CHAIN_*): the workgroup's cotangent scale cs, sized by the larger of the primal
    // cotangent and 2^-CHAIN_Q_OVER_D of the tangent one; the work item is walked again if a split left the format
    float cs = 1.f, amax = 0.f, redo_amax = 0.f;
    int prov = 1;
    for (int attempt = 0;; ++attempt) {
    if (attempt > 0) {
        // the work item again; the third time with the direction 2^CHAIN_V_RETRY lower (its planes were made by k_wb_planes: every
        // fragment is brought down on its way into the registers, wb_fix; the float32 copies here); the first round's observations
        // and the rings' first steps anew
        const int tid = threadIdx.x, lane = tid & 63;
        float *vb1s = sm + L.vb1, *vb2s = sm + L.vb2, *vb3s = sm + L.vb3, *vls = sm + L.vls;
        vfix = attempt >= 2 ? pow2f(-CHAIN_V_RETRY) : 1.f;      // (the second walk keeps the direction's scale, the third lowers it too)
        vs = vs0 * vfix;
        ivs = 1.f / vs;
        klv = a.kl_weight * vs;
        __syncthreads();
        wb_load_w3f(v3f, vv + oW3, A, lane, w, vs);
        for (int e = tid; e < H; e += 256) {
            vb1s[e] = vs * vv[ob1 + e];
            vb2s[e] = vs * vv[ob2 + e];
        }
        if (tid < 16) {
            const float sr = (tid < A) ? th[oS + tid] : 0.f;
            const bool clipped = a.clip_log_std && (sr < a.min_log_std);
            vls[tid] = (tid < A && !clipped) ? vs * vv[oS + tid] : 0.f;
            vb3s[tid] = (tid < A) ? vs * vv[ob3 + tid] : 0.f;
        }
        wb_xreq<NKO>(X, a.obs, wk.row_begin, wk.row_end - wk.row_begin, O, tid);
        wb_stream_begin(GT, PT, NKO, w, lane);
        wb_stream_begin(GV, PV, NKO, w, lane, vfix);
    }
#pragma unroll
    for (int m = 0; m < NXB; ++m) aw1[m] = wb_zero16();
#pragma unroll
    for (int m = 0; m < 4; ++m) aw2[m] = wb_zero16();
    aw3[0] = aw3[1] = zero4();
    klsum = outs = outb3 = 0.f;
    cs = 1.f;
    amax = 0.f;
    prov = 1;
    if (PROMP_NT == 2 && attempt > 0 && redo_amax > 0.f && redo_amax < 3.0e38f) {
        int k = scale_exp(redo_amax, CHAIN_CT_REDO - (attempt - 1) * CHAIN_CT_RETRY);
        k = k < -100 ? -100 : k > 100 ? 100 : k;
        cs = pow2f(k);
        prov = 0;
    }

    int rix = 0;
    for (int base = wk.row_begin; base < wk.row_end; base += R) {
        const int nrows = (wk.row_end - base) < R ? (wk.row_end - base) : R;
        const int zr = opaque_zero();
        const int tid = threadIdx.x + zr, lane = tid & 63, j = lane & 31, h = lane >> 5;
        float* const smz = sm + zr;
        float *Xs = smz + L.x0, *H1s = smz + L.h1, *H2s = smz + L.h2, *RH1s = smz + L.rh1, *RH2s = smz + L.rh2, *Mp = smz + L.mp,
              *Mp2 = smz + L.mp2, *Mss = smz + L.ms, *Ms2s = smz + L.ms2, *Dm = smz + L.dm, *Dm2 = smz + L.dm2, *b1s = smz + L.b1,
              *b2s = smz + L.b2, *b3s = smz + L.b3, *vb1s = smz + L.vb1, *vb2s = smz + L.vb2, *vb3s = smz + L.vb3, *lss = smz + L.ls,
              *ess = smz + L.es, *sn2s = smz + L.sn2, *vls = smz + L.vls, *D1p = smz + L.d1 + w * 1024;
        const unsigned *PTz = PT + zr, *PVz = PV + zr;
        WB_STAMP(0);
        __syncthreads();                       // the previous round is done with every tile
        WB_STAMP(20);
        // ---- this round's observations (requested a round ago) -> the tile; the next round's are requested
        wb_xput<NKO>(Xs, X, nrows, O, OC, tid, xs);
        WB_STAMP(21);
        wb_xreq<NKO>(X, a.obs, (long long)base + R, wk.row_end - base - R, O, tid);
        WB_STAMP(22);
        // ---- the loss level's row data: one (row, action) pair per thread
        const int erow = tid >> 3, eq = tid & 7;
        const bool eown = eq < A, rvalid = erow < nrows;
        float advn, eac, emo, eso;
        {
            const long long n = (long long)base + (rvalid ? erow : 0);
            const int qq = eown ? eq : 0;
            const float* olsp = a.old_log_std + (a.ls_per_row ? n * A : (long long)task * A);
            advn = a.adv[n];
            eac = a.act[n * A + qq];
            emo = a.old_mean[n * A + qq];
            eso = olsp[qq];
        }
        WB_STAMP(1);
        __syncthreads();
        WB_STAMP(2);
        // ---- layer 1 and its tangent:  Rz1 = vW1^T x + vb1
        f32x16 rh1v;
        {
            // the primal chain first (theta's planes), then the tangent chain (the direction's planes) with the primal's tanh, plane split
            // and stores quad by quad among its products: vector work in the matrix instructions' shadow instead of behind them
            f32x16 h1v;
            f32x16 cz = wb_bias16(b1s, h, w);
            u32x4 fb[PROMP_NT];
            wb_read_b(fb, Xs, j, h, 0);
#pragma unroll
            for (int q = 0; q < NKO; ++q) {
                u32x4 fn[PROMP_NT];
                if (q + 1 < NKO) wb_read_b(fn, Xs, j, h, q + 1);
                sched_fence();
                wb_mma6(cz, GT.r[q % WB_PF], fb);
                wb_ring_next(GT, PTz, NKO, PTz + oC2, 8, w, lane, q);
                sched_fence();
                if (q + 1 < NKO) {
#pragma unroll
                    for (int t = 0; t < PROMP_NT; ++t) fb[t] = fn[t];
                }
            }
            f32x16 cr = wb_bias16(vb1s, h, w);
            wb_read_b(fb, Xs, j, h, 0);
#pragma unroll
            for (int q = 0; q < NKO; ++q) {
                u32x4 fn[PROMP_NT];
                if (q + 1 < NKO) wb_read_b(fn, Xs, j, h, q + 1);
                if (q < 4) {
                    const int g = q;
                    f32x4 hq;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        hq[i] = fast_tanh(cz[4 * g + i]);
                        h1v[4 * g + i] = hq[i];
                    }
                    wb_store_quad(H1s, hq[0], hq[1], hq[2], hq[3], j, h, w, g);
                    sts4(D1p + 256 * g + 4 * lane, hq);
                }
                wb_mma6(cr, GV.r[q % WB_PF], fb);
                wb_ring_next(GV, PVz, NKO, PVz + oC2, 8, w, lane, q, vfix);
                sched_fence();
                if (q + 1 < NKO) {
#pragma unroll
                    for (int t = 0; t < PROMP_NT; ++t) fb[t] = fn[t];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) rh1v[r] = (1.f - h1v[r] * h1v[r]) * cr[r];
            wb_store_own(RH1s, rh1v, j, h, w);
        }
        sched_fence();
        WB_STAMP(3);
        __syncthreads();
        WB_STAMP(4);
        // ---- layer 2 and its tangent:  Rz2 = vW2^T h1 + W2^T Rh1 + vb2
        f32x16 h2v, rh2v;
        {
            f32x16 cz = wb_bias16(b2s, h, w), cr = wb_bias16(vb2s, h, w);
            u32x4 fb[PROMP_NT], fr[PROMP_NT];
            wb_read_b(fb, H1s, j, h, 0);
            wb_read_b(fr, RH1s, j, h, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                u32x4 nb[PROMP_NT], nr[PROMP_NT];
                if (q + 1 < 8) {
                    wb_read_b(nb, H1s, j, h, q + 1);
                    wb_read_b(nr, RH1s, j, h, q + 1);
                }
                sched_fence();
                wb_mma6_two(cz, cr, GT.r[q % WB_PF], GV.r[q % WB_PF], fb);
                wb_mma6(cr, GT.r[q % WB_PF], fr);
                wb_ring_next(GT, PTz + oC2, 8, PTz + oR2, 8, w, lane, q);
                wb_ring_next(GV, PVz + oC2, 8, PVz + oR2, 8, w, lane, q, vfix);
                sched_fence();
                if (q + 1 < 8) {
#pragma unroll
                    for (int t = 0; t < PROMP_NT; ++t) {
                        fb[t] = nb[t];
                        fr[t] = nr[t];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float hv = fast_tanh(cz[r]);
                h2v[r] = hv;
                rh2v[r] = (1.f - hv * hv) * cr[r];
            }
            wb_store_own(H2s, h2v, j, h, w);
            wb_store_own(RH2s, rh2v, j, h, w);
        }
        wave_sync();
        sched_fence();
        WB_STAMP(5);
        // ---- output layer and its tangent, this wave's 32 units of the contraction:  Rmu = vW3^T h2 + W3^T Rh2 + vb3
        {
            const int j16 = lane & 15, g4 = lane >> 4;
            u32x4 w3p[PROMP_NT], v3p[PROMP_NT];
            wb_gload(w3p, PTz + oW3P, 1, w, 0, lane);
            wb_gload(v3p, PVz + oW3P, 1, w, 0, lane);
            wb_fix(v3p, vfix);
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const int s = 16 * sb + j16;
                u32x4 fb[PROMP_NT], fr[PROMP_NT];
                const int off = wb_chunk(s, 8 * w + 2 * g4);
#pragma unroll
                for (int t = 0; t < PROMP_NT; ++t) {
                    fb[t] = wb_lds4(H2s + t * WB_PLANE + off);
                    fr[t] = wb_lds4(RH2s + t * WB_PLANE + off);
                }
                f32x4 pm = zero4(), pr = zero4();
                wb_mma6_16(pm, w3p, fb);
                wb_mma6_16(pr, v3p, fb);
                wb_mma6_16(pr, w3p, fr);
                if (g4 < 2) {
                    sts4(Mp + (w * R + s) * 8 + 4 * g4, pm);
                    sts4(Mp2 + (w * R + s) * 8 + 4 * g4, pr);
                }
            }
        }
        WB_STAMP(6);
        __syncthreads();
        WB_STAMP(7);
        // ---- loss-level R-operator (the arithmetic of k_wide_hvp, one (row, action) pair per thread)
        {
            float dlp = 0.f, Rlp = 0.f, kl = 0.f;
            float z = 0.f, ee = 0.f, Rmu = 0.f, dklm = 0.f, dkls = 0.f, Rs = 0.f, kRdm = 0.f, kRds = 0.f;
            if (eown) {
                const float mu = b3s[eq] + ((Mp[(0 * R + erow) * 8 + eq] + Mp[(1 * R + erow) * 8 + eq]) + (Mp[(2 * R + erow) * 8 + eq] + Mp[(3 * R + erow) * 8 + eq]));
                Rmu = vb3s[eq] + ((Mp2[(0 * R + erow) * 8 + eq] + Mp2[(1 * R + erow) * 8 + eq]) + (Mp2[(2 * R + erow) * 8 + eq] + Mp2[(3 * R + erow) * 8 + eq]));
                const float s = lss[eq];
                Rs = vls[eq];
                ee = ess[eq];
                z = (eac - mu) * ee;
                const float zo = (eac - emo) * fast_exp(-eso);
                dlp = (eso - s) - 0.5f * (z * z - zo * zo);
                Rlp = z * ee * Rmu + (z * z - 1.f) * Rs;
                const float sn2 = sn2s[eq], num = (emo - mu) * (emo - mu) + fast_exp(2.f * eso) - sn2, den = 2.f * sn2 + 1e-8f;
                const float rden = fast_rcp(den);
                kl = num * rden + s - eso;
                dklm = -2.f * (emo - mu) * rden * invN;
                dkls = ((-2.f * sn2 * den - 4.f * num * sn2) * (rden * rden) + 1.f) * invN;
                // objective = the mean KL itself (LOSS_KL): R{dKL/dmu}, R{dKL/ds} along v (formulas: k_chain_hvp)
                const float D = emo - mu, Pk = sn2 * (den + 2.f * num);
                const float RP = 2.f * sn2 * Rs * (den + 2.f * num) - 4.f * sn2 * D * Rmu;
                kRdm = (2.f * Rmu * rden + 8.f * D * sn2 * Rs * (rden * rden)) * invN;
                kRds = (-2.f * RP + 16.f * Pk * sn2 * Rs * rden) * (rden * rden) * invN;
            }
#pragma unroll
            for (int m = 1; m <= 4; m <<= 1) {
                dlp += shfl_xor_f32(dlp, m);
                Rlp += shfl_xor_f32(Rlp, m);
                kl += shfl_xor_f32(kl, m);
            }
            float c = 0.f, Rc = 0.f, km = 0.f;
            if (rvalid) {
                km = 1.f;
                if (a.loss_kind == LOSS_RATIO) {
                    c = -advn * expf(dlp) * invN;
                    Rc = c * Rlp;
                } else {
                    c = -advn * invN;
                }
                if (eq == 0) klsum += kl * invN;
            }
            const bool klobj = a.loss_kind == LOSS_KL;      // the outputs are MINUS the tangent of the gradient
            float d = 0.f, qm = 0.f;                        // (the tangent quantities at the direction's scale vs)
            if (eown) {
                const float Rz = -Rmu * ee - z * Rs;
                const float Rd = Rc * z * ee + c * (Rz * ee - z * ee * Rs);
                const float Rds = Rc * (z * z - 1.f) + 2.f * c * z * Rz;
                d = klobj ? km * dklm : c * z * ee;
                qm = klobj ? -km * kRdm : km * (-Rd + klv * dklm);
                outs += klobj ? -km * kRds : km * (-Rds + klv * dkls);
                outb3 += qm;
            }
            if (PROMP_NT == 2) {
                const float am = fmaxf(fabsf(d), (1.f / (float)(1 << CHAIN_Q_OVER_D)) * fabsf(qm));
                amax = fmaxf(amax, am);
                if (prov) {          // (the same in every thread) no round has had a cotangent yet: this one sets the scale
                    const float m = wave_absmax_f32(am);
                    if (lane == 0) red[w] = m;
                    __syncthreads();
                    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
                    const bool okm = mx > 0.f && mx < 3.0e38f;
                    int k = scale_exp(okm ? mx : invN, okm ? CHAIN_CT_TARGET : -4);
                    k = k < -100 ? -100 : k > 100 ? 100 : k;
                    cs = pow2f(k);
                    prov = okm ? 0 : 1;
                }
                d *= cs;
                qm *= cs;
            }
            if (eown) {
                Mss[erow * MS + eq] = -d;
                Ms2s[erow * MS + eq] = qm;
                unsigned t3[PROMP_NT];
                split_pair<PROMP_NT>(-d, qm, t3);
                unsigned short* Dh = (unsigned short*)Dm;
                unsigned short* Dh2 = (unsigned short*)Dm2;
#pragma unroll
                for (int t = 0; t < PROMP_NT; ++t) {
                    Dh[2 * (t * WB_DMPL + erow * WB_DMROW) + eq] = (unsigned short)(t3[t] & 0xFFFFu);
                    Dh2[2 * (t * WB_DMPL + erow * WB_DMROW) + eq] = (unsigned short)(t3[t] >> 16);
                }
            }
        }
        WB_STAMP(8);
        __syncthreads();
        WB_STAMP(9);
        // ---- out_W3 rows 32 w .. += Rh2^T nd + h2^T qmu;  ndZ2 = (W3 nd^T)(1 - h2^2);  qZ2 = (W3 qmu^T + vW3 nd^T)(1 - h2^2) - 2 (W3 nd^T) h2 Rh2
        {
            const int p = lane & 15, g4 = lane >> 4;
            u32x4 fan[PROMP_NT], faq[PROMP_NT];
            {
                const int o0 = (8 * g4 + (p >> 2)) * WB_DMROW + 2 * (p & 3), o1 = o0 + 4 * WB_DMROW;
#pragma unroll
                for (int t = 0; t < PROMP_NT; ++t) {
                    fan[t] = join_w2(lds_tr16(Dm + t * WB_DMPL + o0), lds_tr16(Dm + t * WB_DMPL + o1));
                    faq[t] = join_w2(lds_tr16(Dm2 + t * WB_DMPL + o0), lds_tr16(Dm2 + t * WB_DMPL + o1));
                }
            }
#pragma unroll
            for (int ub = 0; ub < 2; ++ub) {
                u32x4 fbr[PROMP_NT], fbh[PROMP_NT];
                wb_read_tr16(fbr, RH2s, lane, 2 * w + ub);
                wb_read_tr16(fbh, H2s, lane, 2 * w + ub);
                wb_mma6_16(aw3[ub], fan, fbr);
                wb_mma6_16(aw3[ub], faq, fbh);
            }
            f32x16 dn = wb_zero16(), qh = wb_zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float bn = Mss[j * MS + 2 * kk + h], bq = Ms2s[j * MS + 2 * kk + h];
                dn = mfma32(w3f[kk], bn, dn);
                qh = mfma32(w3f[kk], bq, qh);
                qh = mfma32(v3f[kk], bn, qh);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d2 = 1.f - h2v[r] * h2v[r];
                const float qz = qh[r] * d2 - 2.f * dn[r] * h2v[r] * rh2v[r];
                dn[r] *= d2;
                qh[r] = qz;
            }
            wave_sync();                       // this wave's own reads of its H2 / Rh2 columns precede the overwrite
            wb_store_own(H2s, dn, j, h, w);
            wb_store_own(RH2s, qh, j, h, w);
            wave_sync();
            // out_b2: a row of ones (action slot 15) against the wave's own qZ2 columns
            u32x4 ones;
#pragma unroll
            for (int i = 0; i < 4; ++i) ones[i] = (p == 15) ? WB_ONE2 : 0u;
#pragma unroll
            for (int ub = 0; ub < 2; ++ub) {
                u32x4 fb[PROMP_NT];
                wb_read_tr16(fb, RH2s, lane, 2 * w + ub);
#pragma unroll
                for (int t = PROMP_NT - 1; t >= 0; --t) aw3[ub] = mfma16_sw<PROMP_NT>(ones, fb[t], aw3[ub]);
            }
        }
        sched_fence();
        WB_STAMP(10);
        __syncthreads();
        WB_STAMP(11);
        // ---- out_W2 columns 32 w .. += Rh1^T ndZ2 + h1^T qZ2 (one block and sample half per K step), under the backward products
        //      a = W2 ndZ2^T,  aq = W2 qZ2^T + vW2 ndZ2^T  for the own units
        f32x16 qz1;
        {
            f32x16 ad = wb_zero16(), aq = wb_zero16();
            u32x4 fd[PROMP_NT], fq[PROMP_NT], fzn[PROMP_NT], fzq[PROMP_NT];
            wb_read_b(fd, H2s, j, h, 0);
            wb_read_b(fq, RH2s, j, h, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                u32x4 nd[PROMP_NT], nq[PROMP_NT];
                if (q + 1 < 8) {
                    wb_read_b(nd, H2s, j, h, q + 1);
                    wb_read_b(nq, RH2s, j, h, q + 1);
                }
                const int t = q >> 2, m = q & 3;
                if (m == 0) {
                    wb_read_tr(fzn, H2s, lane, w, t);
                    wb_read_tr(fzq, RH2s, lane, w, t);
                }
                u32x4 far[PROMP_NT], fah[PROMP_NT];
                wb_read_tr(far, RH1s, lane, m, t);
                wb_read_tr(fah, H1s, lane, m, t);
                sched_fence();
                wb_mma6_ab(ad, aq, GT.r[q % WB_PF], fd, fq);
                wb_mma6_x2(aq, GV.r[q % WB_PF], fd, aw2[m], far, fzn);
                wb_mma6(aw2[m], fah, fzq);
                wb_ring_next(GT, PTz + oR2, 8, PTz, NKO, w, lane, q);
                wb_ring_next(GV, PVz + oR2, 8, PVz, NKO, w, lane, q, vfix);
                sched_fence();
                if (q + 1 < 8) {
#pragma unroll
                    for (int tt = 0; tt < PROMP_NT; ++tt) {
                        fd[tt] = nd[tt];
                        fq[tt] = nq[tt];
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 h1q = lds4(D1p + 256 * g + 4 * lane);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * g + i;
                    qz1[r] = aq[r] * (1.f - h1q[i] * h1q[i]) - 2.f * ad[r] * h1q[i] * rh1v[r];
                }
            }
        }
        sched_fence();
        WB_STAMP(12);
        __syncthreads();                       // every wave has read all of ndZ2 / qZ2 (and h1 / Rh1) before qZ1 takes the ndZ2 columns' place
        WB_STAMP(13);
        wb_store_own(H2s, qz1, j, h, w);
        wave_sync();
        // ---- out_W1 columns 32 w .. += x^T qZ1 (row OC: out_b1)
        {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4 fz[PROMP_NT];
                wb_read_tr(fz, H2s, lane, w, t);
                if (NXB == 1) {
                    u32x4 fa[PROMP_NT];
                    wb_read_tr(fa, Xs, lane, 0, t);
                    wb_mma6(aw1[0], fa, fz);
                } else {
#pragma unroll
                    for (int m = 0; m + 1 < NXB; m += 2) {
                        u32x4 fa0[PROMP_NT], fa1[PROMP_NT];
                        wb_read_tr(fa0, Xs, lane, m, t);
                        wb_read_tr(fa1, Xs, lane, m + 1, t);
                        wb_mma6_two(aw1[m], aw1[m + 1], fa0, fa1, fz);
                    }
                }
            }
        }
        WB_STAMP(14);
        ++rix;
    }
    // FP16 split: every split of this kernel ends in the hidden_0 kernel sums (x * 0 is 0 for finite x only)
    if (PROMP_NT != 2 || attempt + 1 >= CHAIN_ATTEMPTS) break;
    {
        float chk = 0.f;
#pragma unroll
        for (int m = 0; m < NXB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) chk = __builtin_fmaf(aw1[m][r], 0.f, chk);
        const float wm = wave_absmax_f32(amax), wbad = wave_any(chk != chk) ? 1.f : 0.f;
        const int lane = threadIdx.x & 63;
        __syncthreads();
        if (lane == 0) {
            red[w] = wm;
            red[4 + w] = wbad;
        }
        __syncthreads();
        redo_amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        // (the mean cotangents' planes feed the output kernel's sums only: their largest values at the scale -- amax holds the larger of
        //  |d| and 2^-CHAIN_Q_OVER_D |q|)
        const bool bad = (red[4] + red[5] + red[6] + red[7]) > 0.f || !(redo_amax * cs * (float)(1 << CHAIN_Q_OVER_D) <= 65504.f);
        if (!bad) break;
        if (threadIdx.x == 0) atomic_add_agent(a.split_events + 1, 1);
    }
    }
    const float us = PROMP_NT == 2 ? 1.f / (cs * vs) : 1.f;      // what the tangent cotangents' sums carry
    outs *= ivs;
    outb3 *= ivs;

    float* P = a.partials + (long long)wi * a.partial_stride;
    const int tidt = threadIdx.x + opaque_zero(), lane = tidt & 63, j = lane & 31, h = lane >> 5, tid = tidt;
#pragma unroll
    for (int m = 8; m <= 32; m <<= 1) {
        outs += shfl_xor_f32(outs, m);
        outb3 += shfl_xor_f32(outb3, m);
        klsum += shfl_xor_f32(klsum, m);
    }
    __syncthreads();
    if (lane < 8) {
        float* rw = red + 16 * w;
        rw[lane] = outs;
        rw[8 + lane] = outb3;
    }
    float* sc = Mp;
    if (lane == 0) sc[w] = klsum;
    __syncthreads();
    if (tid == 0) {
        P[NP] = 0.f;
        P[NP + 1] = (sc[0] + sc[1]) + (sc[2] + sc[3]);
    }
    if (tid < 16) {
        const int aidx = tid & 7, which = tid >> 3;
        const float t = (red[8 * which + aidx] + red[16 + 8 * which + aidx]) + (red[32 + 8 * which + aidx] + red[48 + 8 * which + aidx]);
        if (aidx < A) {
            if (which == 0) P[oS + aidx] = t * lmask[aidx];
            else P[ob3 + aidx] = t;
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) P[oW2 + (32 * m + 8 * (r >> 2) + 4 * h + (r & 3)) * H + 32 * w + j] = aw2[m][r] * us;
#pragma unroll
    for (int m = 0; m < NXB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * m + 8 * (r >> 2) + 4 * h + (r & 3);
            if (row < O) P[row * H + 32 * w + j] = aw1[m][r] * (us * w1u);
            else if (row == OC) P[ob1 + 32 * w + j] = aw1[m][r] * us;
        }
    {
        const int j16 = lane & 15, g4 = lane >> 4;
#pragma unroll
        for (int ub = 0; ub < 2; ++ub)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int aa = 4 * g4 + r, unit = 32 * w + 16 * ub + j16;
                if (aa < A) P[oW3 + unit * A + aa] = aw3[ub][r] * us;
                else if (aa == 15) P[ob2 + unit] = aw3[ub][r] * us;
            }
    }
}
