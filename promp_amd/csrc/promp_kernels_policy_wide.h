// promp_kernels_policy_wide.h -- the policy passes for obs_dim up to 128 and hidden width 64 or 128
// (BASELINE config 4: AntRandDirec, obs 111, act 8, 2x128 tanh MLP; also Ant with the reference's default 2x64).
//
// Same mathematics, arguments and partial-sum layout as k_pass / k_chain_hvp (promp_kernels_pass.h, promp_kernels_chain.h), different
// decomposition.  At H = 128 one wave can no longer hold a whole hidden_1 gradient (128x128 = 256 accumulator
// registers per lane), and theta (137 KB) no longer fits next to the activation tiles in the 160 KB LDS.  So the
// workgroup turns cooperative:
//
//   * H/16 waves (8 at H = 128); wave w OWNS the 16-column block w of every hidden layer: it produces those columns of H1 / H2 / dZ,
//     and accumulates exactly those columns of the hidden_0 / hidden_1 kernel gradients (its 16 rows of the output
//     kernel) over ALL rows the workgroup walks -- no cross-wave reduction at the end, every partial entry is
//     written once, by one lane, in a fixed order.
//   * rounds of R rows (64 for fwd/bwd, 32 for the R-operator pass) live in LDS: X, H1, H2 (+ tangents), the
//     mean tile.  MFMA A operands are LDS reads; workgroup barriers separate the layer phases.
//   * MFMA B operands (weights) never touch LDS: each lane keeps the slice of W1 / W2 it multiplies with in
//     registers (fwd/bwd: loaded once per kernel; R-operator pass: theta and v slices re-read from L2 each phase).
//     Only the tiny output kernel (128 x 8) sits in LDS.
//   * k mapping of a step (the four k values lane groups kk = 0..3 contract) is chosen per GEMM so that the two
//     lane groups of a 32-lane LDS access are 16 k (or 16 rows) apart: conflict-free with the odd tile strides.
#pragma once
#include "promp_kernels_policy.h"

#define WIDE_MS 17
#define WIDE_W3S 17

struct LdsWide {
    int x, h1, h2, rh1, rh2, ms, ms2, w3, w3t, vw3, vw3t, b1, b2, b3, vb1, vb2, vb3, ls, lmask, es, sn2, vls, red;
    int XS, total;
};

// RB = 16-row blocks per round; NOB = 16-column blocks of the (zero-padded) observation
PROMP_HD LdsWide make_layout_wide(int H, int RB, int NOB, bool hvp) {
    const int WIDE_H = H, WIDE_HS = H + 1;
    LdsWide L;
    int o = 0;
    const int R = 16 * RB;
    L.XS = 16 * NOB + 1;
#define WIDE_TAKE(field, n) \
    L.field = o;            \
    o += ((n) + 3) & ~3
    WIDE_TAKE(x, R * L.XS);
    WIDE_TAKE(h1, R * WIDE_HS);
    WIDE_TAKE(h2, R * WIDE_HS);
    WIDE_TAKE(ms, R * WIDE_MS);
    WIDE_TAKE(w3, WIDE_H * WIDE_W3S);
    WIDE_TAKE(w3t, 8 * WIDE_HS);
    WIDE_TAKE(b1, WIDE_H);
    WIDE_TAKE(b2, WIDE_H);
    WIDE_TAKE(b3, 16);
    WIDE_TAKE(ls, 16);
    WIDE_TAKE(lmask, 16);
    WIDE_TAKE(es, 16);
    WIDE_TAKE(sn2, 16);
    WIDE_TAKE(red, 64);
    L.rh1 = L.rh2 = L.ms2 = L.vw3 = L.vw3t = L.vb1 = L.vb2 = L.vb3 = L.vls = 0;
    if (hvp) {
        WIDE_TAKE(rh1, R * WIDE_HS);
        WIDE_TAKE(rh2, R * WIDE_HS);
        WIDE_TAKE(ms2, R * WIDE_MS);
        WIDE_TAKE(vw3, WIDE_H * WIDE_W3S);
        WIDE_TAKE(vw3t, 8 * WIDE_HS);
        WIDE_TAKE(vb1, WIDE_H);
        WIDE_TAKE(vb2, WIDE_H);
        WIDE_TAKE(vb3, 16);
        WIDE_TAKE(vls, 16);
    }
#undef WIDE_TAKE
    L.total = o;
    return L;
}

// k value lane group kk contracts at step s of a K-deep GEMM (K = 32, 64 or 128)
template <int K>
PROMP_DEV int wide_kmap(int s, int kk) {
    if (K >= 64) return (s & 15) + 16 * kk + 64 * (s >> 4);
    return s + (K / 4) * kk;
}
// row of an R-row round lane group kk contracts at step s of a gradient GEMM (K = rows)
template <int RB>
PROMP_DEV int wide_rowmap(int s, int kk) {
    if (RB == 4) return s + 16 * kk;                       // 16 steps
    return s + 16 * (kk & 1) + 8 * (kk >> 1);              // RB == 2: 8 steps
}

// this lane's slice of a [K][H] kernel as the B operand of  out[:, 16w..16w+15] = act[:, K] * W :  W[kmap(s,kk)][16w+i16]
// (32-bit unsigned element offsets from the wave-uniform base: one address register + immediates per group of steps)
template <int K, int WIDE_H>
PROMP_DEV void wide_load_cols(float (&r)[K / 4], const float* W, int krows, int col, int kk, float sgn) {
#pragma unroll
    for (int s = 0; s < K / 4; ++s) {
        const int k = wide_kmap<K>(s, kk);
        // rows k >= krows (the zero padding of the observation) are read anyway -- K * H <= Theta for every supported
        // shape, so the address is inside the parameter vector -- and discarded: no data-dependent address
        const float v = (W + (wide_kmap<K>(0, kk) * WIDE_H + col))[(wide_kmap<K>(s, 0)) * WIDE_H];   // lane base + constant
        r[s] = (k < krows) ? sgn * v : 0.f;
    }
}
// ... and as the B operand of  out[:, 16w..] = dZ[:, H] * W^T :  W[16w+i16][kmap(s,kk)]
template <int WIDE_H>
PROMP_DEV void wide_load_rows(float (&r)[WIDE_H / 4], const float* W, int row, int kk, float sgn) {
#pragma unroll
    for (int s = 0; s < WIDE_H / 4; ++s) r[s] = sgn * (W + (row * WIDE_H + wide_kmap<WIDE_H>(0, kk)))[wide_kmap<WIDE_H>(s, 0)];
}

// acc[rb] += act[16rb + i16][kmap(s,kk)] * breg[s]   (activation tile in LDS, weights in registers)
template <int RB, int K>
PROMP_DEV void wide_gemm_reg(f32x4 (&acc)[RB], const float* act, int stride, const float (&breg)[K / 4], int i16, int kk) {
#pragma unroll
    for (int s = 0; s < K / 4; ++s) {
        const int k = wide_kmap<K>(s, kk);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma16(act[(16 * rb + i16) * stride + k], breg[s], acc[rb]);
    }
}

// stage the output kernel (and its transpose), biases and log-std terms of one parameter vector
template <int WIDE_H>
PROMP_DEV void wide_stage_head(float* W3s, float* W3Ts, float* b1s, float* b2s, float* b3s, const float* src, int O, int A,
                               int tid, int nthreads) {
    constexpr int WIDE_HS = WIDE_H + 1;
    const int ob1 = O * WIDE_H, oW2 = ob1 + WIDE_H, ob2 = oW2 + WIDE_H * WIDE_H, oW3 = ob2 + WIDE_H, ob3 = oW3 + WIDE_H * A;
    for (int e = tid; e < WIDE_H * 16; e += nthreads) {
        const int k = e >> 4, j = e & 15;
        W3s[k * WIDE_W3S + j] = (j < A) ? src[oW3 + k * A + j] : 0.f;
    }
    for (int e = tid; e < 8 * WIDE_H; e += nthreads) {
        const int aa = e / WIDE_H, k = e - aa * WIDE_H;
        W3Ts[aa * WIDE_HS + k] = (aa < A) ? src[oW3 + k * A + aa] : 0.f;
    }
    for (int e = tid; e < WIDE_H; e += nthreads) {
        b1s[e] = src[ob1 + e];
        b2s[e] = src[ob2 + e];
    }
    if (tid < 16) b3s[tid] = (tid < A) ? src[ob3 + tid] : 0.f;
}

// one round's observations -> LDS tile (rows >= nrows zero, columns >= O stay zero from the initial fill)
PROMP_DEV void wide_load_x(float* Xs, int XS, const float* obs, long long row0, int nrows, int R, int O, int tid, int nthreads) {
    const float rO = 1.0f / (float)O;
    const int lim = nrows * O;
    for (int e = tid; e < R * O; e += nthreads) {
        const int r = (int)(((float)e + 0.5f) * rO);
        const float x = obs[row0 * O + (e < lim ? e : 0)];
        Xs[r * XS + (e - r * O)] = (e < lim) ? x : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// k_wide_fwd_bwd: objective (+ gradient) of one policy pass; PassArgs / partial layout of k_pass.
// grid = work items (table 0), block = 512, one workgroup per CU.
// ---------------------------------------------------------------------------------------------
template <int H, int NOB, bool BWD>
__global__ void __launch_bounds__(4 * H, 2) k_wide_fwd_bwd(PassArgs a) {
    constexpr int RB = 4, R = 16 * RB, HS = H + 1, MS = WIDE_MS, W3S = WIDE_W3S, KO = 16 * NOB, NT = 4 * H;
    constexpr int WIDE_NC = H / 16, WIDE_NS = H / 4;
    PROMP_SMEM_DECL;
    float* sm = (float*)PROMP_SMEM_PTR;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i16_ = lane & 15, kk_ = lane >> 4;
    const int wi = xcd_item(blockIdx.x, gridDim.x);       // which work item (and partial row) this workgroup takes
    const WorkItem wk = a.work[wi];
    const int task = wk.task;
    const int O = a.O, A = a.A;
    const int ob1 = O * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW3 = ob2 + H, ob3 = oW3 + H * A, oS = ob3 + A, NP = oS + A;
    const LdsWide L = make_layout_wide(H, RB, NOB, false);
    const int XS = L.XS;
    float *Xs = sm + L.x, *W3s = sm + L.w3, *W3Ts = sm + L.w3t, *b1s = sm + L.b1, *b2s = sm + L.b2, *b3s = sm + L.b3,
          *lss = sm + L.ls, *lmask = sm + L.lmask, *ess = sm + L.es, *sn2s = sm + L.sn2, *red = sm + L.red;
    const int ntask = a.task_row_offsets[task + 1] - a.task_row_offsets[task];
    const float invN = 1.0f / (float)ntask;
    const float* th = a.theta + (long long)task * a.theta_task_stride;
    const int col_ = 16 * w + i16_;   // the hidden column this lane owns in every MFMA result

    for (int e = tid; e < R * XS; e += NT) Xs[e] = 0.f;
    wide_stage_head<H>(W3s, W3Ts, b1s, b2s, b3s, th, O, A, tid, NT);
    if (tid < 16) {
        const float sr = (tid < A) ? th[oS + tid] : 0.f;
        const bool clipped = a.clip_log_std && (sr < a.min_log_std);   // tf.maximum: gradient iff var >= min
        const float s = clipped ? a.min_log_std : sr;
        lss[tid] = s;
        lmask[tid] = clipped ? 0.f : 1.f;
        ess[tid] = expf(-s);
        sn2s[tid] = expf(2.f * s);
    }
    // weight slices of this lane, resident for the whole kernel
    float w1r[KO / 4], w2c[WIDE_NS], w2r[WIDE_NS];
    wide_load_cols<KO, H>(w1r, th, O, col_, kk_, 1.f);
    wide_load_cols<H, H>(w2c, th + oW2, H, col_, kk_, 1.f);
    if (BWD) wide_load_rows<H>(w2r, th + oW2, col_, kk_, 1.f);

    // gradient slices: hidden_0 kernel [16 ob + ..][col], hidden_1 kernel [16 i + ..][col], output kernel rows 16w..
    f32x4 aw1[NOB], aw2[WIDE_NC], aw3 = zero4();
#pragma unroll
    for (int i = 0; i < NOB; ++i) aw1[i] = zero4();
#pragma unroll
    for (int i = 0; i < WIDE_NC; ++i) aw2[i] = zero4();
    float gb1 = 0.f, gb2 = 0.f;
    float loss = 0.f, klsum = 0.f, gs0 = 0.f, gs1 = 0.f, gb30 = 0.f, gb31 = 0.f;

    // epilogue role (threads 0..4R-1): 4 lanes per row, actions {q, q+4}

    for (int base = wk.row_begin; base < wk.row_end; base += R) {
        const int nrows = (wk.row_end - base) < R ? (wk.row_end - base) : R;
        const int zr = opaque_zero();               // loop-variant lane indices and LDS base (see opaque_zero)
        // (recomputed from the thread index every round, not kept: three lane constants fewer alive across the round)
        const int i16 = ((tid + zr) & 15), kk = (((tid + zr) & 63) >> 4), col = 16 * ((tid + zr) >> 6) + i16;
        // (the epilogue role too -- threads 0..4R-1: 4 lanes per row, actions {q, q+4} -- and the thread index the row loads are
        //  spread by: kept across the round they are spilled, and a scratch reload in front of the observation loads waits for
        //  every load issued before it)
        const int tidz = tid + zr;
        const int erow = tidz >> 2, q = tidz & 3;
        const bool epi = tidz < 4 * R;
        const bool own0 = q < A, own1 = (q + 4) < A;
        const int q0 = own0 ? q : 0, q1 = own1 ? q + 4 : 0;
        float* const smz = sm + zr;
        float *Xs = smz + L.x, *H1s = smz + L.h1, *H2s = smz + L.h2, *Mss = smz + L.ms, *W3s = smz + L.w3, *W3Ts = smz + L.w3t,
              *b1s = smz + L.b1, *b2s = smz + L.b2, *b3s = smz + L.b3, *lss = smz + L.ls, *ess = smz + L.es, *sn2s = smz + L.sn2;
        __syncthreads();                       // the previous round is done with X / H1
        wide_load_x(Xs, XS, a.obs, base, nrows, R, O, tidz, NT);
        __syncthreads();
        // ---- layer 1: H1 = tanh(X W1 + b1)
        {
            f32x4 acc[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = splat4(b1s[col]);
            wide_gemm_reg<RB, KO>(acc, Xs, XS, w1r, i16, kk);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) H1s[(16 * rb + 4 * kk + r) * HS + col] = fast_tanh(acc[rb][r]);
        }
        __syncthreads();
        // ---- layer 2
        {
            f32x4 acc[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = splat4(b2s[col]);
            wide_gemm_reg<RB, H>(acc, H1s, HS, w2c, i16, kk);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) H2s[(16 * rb + 4 * kk + r) * HS + col] = fast_tanh(acc[rb][r]);
        }
        __syncthreads();
        // ---- output layer (16 padded columns): wave rb computes row block rb
        if (w < RB) {
            f32x4 acc = splat4(b3s[i16]);
#pragma unroll
            for (int s = 0; s < WIDE_NS; ++s) {
                const int k = wide_kmap<H>(s, kk);
                acc = mfma16(H2s[(16 * w + i16) * HS + k], W3s[k * W3S + i16], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Mss[(16 * w + 4 * kk + r) * MS + i16] = acc[r];
        }
        __syncthreads();
        // ---- distribution + objective epilogue (same arithmetic as k_pass)
        if (epi) {
            const bool rvalid = erow < nrows;
            const long long n = (long long)base + (rvalid ? erow : 0);
            const float* olsp = a.old_log_std + (a.ls_per_row ? n * A : (long long)task * A);
            const float advn = rvalid ? a.adv[n] : 0.f;
            const float ac0 = a.act[n * A + q0], ac1 = a.act[n * A + q1];
            const float mo0 = a.old_mean[n * A + q0], mo1 = a.old_mean[n * A + q1];
            const float so0 = olsp[q0], so1 = olsp[q1];
            float dlp = 0.f, sumz2 = 0.f, sums = 0.f, kl = 0.f;
            float z0 = 0.f, z1 = 0.f, e0 = 0.f, e1 = 0.f, dklm0 = 0.f, dklm1 = 0.f, dkls0 = 0.f, dkls1 = 0.f;
            if (own0) {
                const float s = lss[q], mu = Mss[erow * MS + q];
                e0 = ess[q];
                z0 = (ac0 - mu) * e0;
                const float zo = (ac0 - mo0) * fast_exp(-so0);
                dlp += (so0 - s) - 0.5f * (z0 * z0 - zo * zo);
                sumz2 += z0 * z0;
                sums += s;
                const float sn2 = sn2s[q], num = (mo0 - mu) * (mo0 - mu) + fast_exp(2.f * so0) - sn2, den = 2.f * sn2 + 1e-8f;
                const float rden = fast_rcp(den);
                kl += num * rden + s - so0;
                dklm0 = -2.f * (mo0 - mu) * rden;
                dkls0 = (-2.f * sn2 * den - 4.f * num * sn2) * (rden * rden) + 1.f;
            }
            if (own1) {
                const float s = lss[q + 4], mu = Mss[erow * MS + q + 4];
                e1 = ess[q + 4];
                z1 = (ac1 - mu) * e1;
                const float zo = (ac1 - mo1) * fast_exp(-so1);
                dlp += (so1 - s) - 0.5f * (z1 * z1 - zo * zo);
                sumz2 += z1 * z1;
                sums += s;
                const float sn2 = sn2s[q + 4], num = (mo1 - mu) * (mo1 - mu) + fast_exp(2.f * so1) - sn2, den = 2.f * sn2 + 1e-8f;
                const float rden = fast_rcp(den);
                kl += num * rden + s - so1;
                dklm1 = -2.f * (mo1 - mu) * rden;
                dkls1 = (-2.f * sn2 * den - 4.f * num * sn2) * (rden * rden) + 1.f;
            }
            dlp += shfl_xor_f32(dlp, 1);  dlp += shfl_xor_f32(dlp, 2);
            sumz2 += shfl_xor_f32(sumz2, 1);  sumz2 += shfl_xor_f32(sumz2, 2);
            sums += shfl_xor_f32(sums, 1);  sums += shfl_xor_f32(sums, 2);
            kl += shfl_xor_f32(kl, 1);  kl += shfl_xor_f32(kl, 2);
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
                if (q == 0) {
                    loss += lrow;
                    klsum += kl * invN;
                }
            }
            if (own0) {
                const float d = c * z0 * e0 + ck * dklm0;
                Mss[erow * MS + q] = d;
                gs0 += c * (z0 * z0 - 1.f) + ck * dkls0;
                gb30 += d;
            }
            if (own1) {
                const float d = c * z1 * e1 + ck * dklm1;
                Mss[erow * MS + q + 4] = d;
                gs1 += c * (z1 * z1 - 1.f) + ck * dkls1;
                gb31 += d;
            }
        }
        if (!BWD) continue;
        __syncthreads();
        // ---- output-kernel gradient rows 16w.. (+=); dZ2 = (dmu W3^T) * (1 - H2^2) in place over this wave's H2 columns
        {
#pragma unroll
            for (int s = 0; s < 4 * RB; ++s) {
                const int row = wide_rowmap<RB>(s, kk);
                aw3 = mfma16(H2s[row * HS + col], Mss[row * MS + i16], aw3);
            }
            f32x4 acc[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = zero4();
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    acc[rb] = mfma16(Mss[(16 * rb + i16) * MS + 4 * s + kk], W3Ts[(4 * s + kk) * HS + col], acc[rb]);
            wave_sync();   // this wave's own reads of its H2 columns (above) precede the overwrite
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = (16 * rb + 4 * kk + r) * HS + col;
                    const float h = H2s[idx];
                    const float d = acc[rb][r] * (1.f - h * h);
                    H2s[idx] = d;
                    gb2 += d;
                }
        }
        __syncthreads();
        // ---- hidden_1 kernel gradient columns 16w.. (+=) and dH1 = dZ2 W2^T for the same columns
        f32x4 dh1[RB];
        {
#pragma unroll
            for (int s = 0; s < 4 * RB; ++s) {
                const int row = wide_rowmap<RB>(s, kk);
                const float b = H2s[row * HS + col];
#pragma unroll
                for (int i = 0; i < WIDE_NC; ++i) aw2[i] = mfma16(H1s[row * HS + 16 * i + i16], b, aw2[i]);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) dh1[rb] = zero4();
            wide_gemm_reg<RB, H>(dh1, H2s, HS, w2r, i16, kk);
        }
        __syncthreads();   // every wave has read all of H1 before the in-place dZ1
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = (16 * rb + 4 * kk + r) * HS + col;
                const float h = H1s[idx];
                const float d = dh1[rb][r] * (1.f - h * h);
                H1s[idx] = d;
                gb1 += d;
            }
        wave_sync();
        // ---- hidden_0 kernel gradient columns 16w.. (+=): needs only this wave's own dZ1 columns
#pragma unroll
        for (int s = 0; s < 4 * RB; ++s) {
            const int row = wide_rowmap<RB>(s, kk);
            const float b = H1s[row * HS + col];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) aw1[ob] = mfma16(Xs[row * XS + 16 * ob + i16], b, aw1[ob]);
        }
    }

    // ---- results: scalars through LDS in wave order, gradient slices straight from their owners ----
    const int tidt = tid + opaque_zero();          // (the tail's lane indices too: recomputed, not reloaded from a spill)
    const int i16 = tidt & 15, kk = (tidt & 63) >> 4, col = 16 * (tidt >> 6) + i16;
    float* P = a.partials + (long long)wi * a.partial_stride;
#pragma unroll
    for (int m = 4; m <= 32; m <<= 1) {
        gs0 += shfl_xor_f32(gs0, m);  gs1 += shfl_xor_f32(gs1, m);  gb30 += shfl_xor_f32(gb30, m);
        gb31 += shfl_xor_f32(gb31, m);  loss += shfl_xor_f32(loss, m);  klsum += shfl_xor_f32(klsum, m);
    }
    __syncthreads();
    if (lane < 4 && w < 4) {       // waves 0..3 ran the epilogue; lane == q
        float* rw = red + 16 * w;
        rw[lane] = gs0;  rw[4 + lane] = gs1;  rw[8 + lane] = gb30;  rw[12 + lane] = gb31;
    }
    __syncthreads();
    float* sc = Xs;                // two scalars per wave
    if (lane == 0) {
        sc[2 * w] = loss;
        sc[2 * w + 1] = klsum;
    }
    __syncthreads();
    if (tid == 0) {
        float l = 0.f, k = 0.f;
        for (int ww = 0; ww < 4; ++ww) {
            l += sc[2 * ww];
            k += sc[2 * ww + 1];
        }
        P[NP] = l;
        P[NP + 1] = k;
    }
    if (!BWD) return;
    if (tid < 16) {
        const int j = tid & 3, which = tid >> 2;     // which: 0 gs(q) 1 gs(q+4) 2 gb3(q) 3 gb3(q+4)
        float t = 0.f;
        for (int ww = 0; ww < 4; ++ww) t += red[16 * ww + 4 * which + j];
        const int aidx = j + ((which & 1) ? 4 : 0);
        if (aidx < A) {
            if (which < 2) P[oS + aidx] = t * lmask[aidx];
            else P[ob3 + aidx] = t;
        }
    }
    gb1 += shfl_xor_f32(gb1, 16);  gb1 += shfl_xor_f32(gb1, 32);
    gb2 += shfl_xor_f32(gb2, 16);  gb2 += shfl_xor_f32(gb2, 32);
    if (kk == 0) {
        P[ob1 + col] = gb1;
        P[ob2 + col] = gb2;
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * ob + 4 * kk + r;
            if (row < O) P[row * H + col] = aw1[ob][r];
        }
#pragma unroll
    for (int i = 0; i < WIDE_NC; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[oW2 + (16 * i + 4 * kk + r) * H + col] = aw2[i][r];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (i16 < A) P[oW3 + (16 * w + 4 * kk + r) * A + i16] = aw3[r];
}

// ---------------------------------------------------------------------------------------------
// k_wide_hvp:  out = -(d^2 L/d theta^2) v + kl_weight * grad KL  for H = 128 (same quantities as k_hvp).
// Rounds of 32 rows; theta and v weight slices are re-read (L2) at the head of each phase.
// grid = work items (table 0), block = 512.
// ---------------------------------------------------------------------------------------------
template <int H, int NOB>
__global__ void __launch_bounds__(4 * H, 2) k_wide_hvp(PassArgs a) {
    constexpr int RB = 2, R = 16 * RB, HS = H + 1, MS = WIDE_MS, W3S = WIDE_W3S, KO = 16 * NOB, NT = 4 * H;
    constexpr int WIDE_NC = H / 16, WIDE_NS = H / 4;
    PROMP_SMEM_DECL;
    float* sm = (float*)PROMP_SMEM_PTR;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i16_ = lane & 15, kk_ = lane >> 4;
    const int wi = xcd_item(blockIdx.x, gridDim.x);       // which work item (and partial row) this workgroup takes
    const WorkItem wk = a.work[wi];
    const int task = wk.task;
    const int O = a.O, A = a.A;
    const int ob1 = O * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW3 = ob2 + H, ob3 = oW3 + H * A, oS = ob3 + A, NP = oS + A;
    const LdsWide L = make_layout_wide(H, RB, NOB, true);
    const int XS = L.XS;
    float *Xs = sm + L.x, *W3s = sm + L.w3, *W3Ts = sm + L.w3t, *vW3s = sm + L.vw3, *vW3Ts = sm + L.vw3t, *b1s = sm + L.b1,
          *b2s = sm + L.b2, *b3s = sm + L.b3, *vb1s = sm + L.vb1, *vb2s = sm + L.vb2, *vb3s = sm + L.vb3, *lss = sm + L.ls,
          *lmask = sm + L.lmask, *ess = sm + L.es, *sn2s = sm + L.sn2, *vls = sm + L.vls, *red = sm + L.red;
    const int ntask = a.task_row_offsets[task + 1] - a.task_row_offsets[task];
    const float invN = 1.0f / (float)ntask;
    const float* th0 = a.theta + (long long)task * a.theta_task_stride;
    const float* v0 = a.vdir + (long long)task * NP;
    const int col_ = 16 * w + i16_;

    for (int e = tid; e < R * XS; e += NT) Xs[e] = 0.f;
    wide_stage_head<H>(W3s, W3Ts, b1s, b2s, b3s, th0, O, A, tid, NT);
    wide_stage_head<H>(vW3s, vW3Ts, vb1s, vb2s, vb3s, v0, O, A, tid, NT);
    if (tid < 16) {
        const float* th = th0;
        const float* v = v0;
        const float sr = (tid < A) ? th[oS + tid] : 0.f;
        const bool clipped = a.clip_log_std && (sr < a.min_log_std);
        const float s = clipped ? a.min_log_std : sr;
        lss[tid] = s;
        lmask[tid] = clipped ? 0.f : 1.f;
        ess[tid] = expf(-s);
        sn2s[tid] = expf(2.f * s);
        vls[tid] = (tid < A && !clipped) ? v[oS + tid] : 0.f;   // R{s} = mask * v_s
    }

    f32x4 aw1[NOB], aw2[WIDE_NC], aw3 = zero4();
#pragma unroll
    for (int i = 0; i < NOB; ++i) aw1[i] = zero4();
#pragma unroll
    for (int i = 0; i < WIDE_NC; ++i) aw2[i] = zero4();
    float ob1acc = 0.f, ob2acc = 0.f;
    float klsum = 0.f, outs0 = 0.f, outs1 = 0.f, outb30 = 0.f, outb31 = 0.f;
    const float klw = a.kl_weight;


    for (int base = wk.row_begin; base < wk.row_end; base += R) {
        const int nrows = (wk.row_end - base) < R ? (wk.row_end - base) : R;
        const int zr = opaque_zero();               // loop-variant lane indices, LDS and parameter bases (see opaque_zero)
        // (recomputed from the thread index every round, not kept: three lane constants fewer alive across the round)
        const int i16 = ((tid + zr) & 15), kk = (((tid + zr) & 63) >> 4), col = 16 * ((tid + zr) >> 6) + i16;
        // (the epilogue role too -- threads 0..4R-1: 4 lanes per row, actions {q, q+4} -- and the thread index the row loads are
        //  spread by: kept across the round they are spilled, and a scratch reload in front of the observation loads waits for
        //  every load issued before it)
        const int tidz = tid + zr;
        const int erow = tidz >> 2, q = tidz & 3;
        const bool epi = tidz < 4 * R;
        const bool own0 = q < A, own1 = (q + 4) < A;
        const int q0 = own0 ? q : 0, q1 = own1 ? q + 4 : 0;
        float* const smz = sm + zr;
        const float *th = th0 + zr, *v = v0 + zr;
        float *Xs = smz + L.x, *H1s = smz + L.h1, *H2s = smz + L.h2, *RH1s = smz + L.rh1, *RH2s = smz + L.rh2, *Mss = smz + L.ms,
              *Ms2s = smz + L.ms2, *W3s = smz + L.w3, *W3Ts = smz + L.w3t, *vW3s = smz + L.vw3, *vW3Ts = smz + L.vw3t,
              *b1s = smz + L.b1, *b2s = smz + L.b2, *b3s = smz + L.b3, *vb1s = smz + L.vb1, *vb2s = smz + L.vb2,
              *vb3s = smz + L.vb3, *lss = smz + L.ls, *ess = smz + L.es, *sn2s = smz + L.sn2, *vls = smz + L.vls;
        __syncthreads();
        wide_load_x(Xs, XS, a.obs, base, nrows, R, O, tidz, NT);
        // ---- layer 1 and its tangent:  Rz1 = X vW1 + vb1
        {
            float wr[KO / 4], vr[KO / 4];
            sched_fence();
            wide_load_cols<KO, H>(wr, th, O, col, kk, 1.f);
            wide_load_cols<KO, H>(vr, v, O, col, kk, 1.f);
            __syncthreads();
            f32x4 az[RB], ar[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                az[rb] = splat4(b1s[col]);
                ar[rb] = splat4(vb1s[col]);
            }
            wide_gemm_reg<RB, KO>(az, Xs, XS, wr, i16, kk);
            wide_gemm_reg<RB, KO>(ar, Xs, XS, vr, i16, kk);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = (16 * rb + 4 * kk + r) * HS + col;
                    const float h = fast_tanh(az[rb][r]);
                    H1s[idx] = h;
                    RH1s[idx] = (1.f - h * h) * ar[rb][r];
                }
        }
        // ---- layer 2 and its tangent:  Rz2 = H1 vW2 + RH1 W2 + vb2
        {
            float wr[WIDE_NS], vr[WIDE_NS];
            sched_fence();
            wide_load_cols<H, H>(wr, th + oW2, H, col, kk, 1.f);
            wide_load_cols<H, H>(vr, v + oW2, H, col, kk, 1.f);
            __syncthreads();
            f32x4 az[RB], ar[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                az[rb] = splat4(b2s[col]);
                ar[rb] = splat4(vb2s[col]);
            }
            wide_gemm_reg<RB, H>(az, H1s, HS, wr, i16, kk);
            wide_gemm_reg<RB, H>(ar, H1s, HS, vr, i16, kk);
            wide_gemm_reg<RB, H>(ar, RH1s, HS, wr, i16, kk);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = (16 * rb + 4 * kk + r) * HS + col;
                    const float h = fast_tanh(az[rb][r]);
                    H2s[idx] = h;
                    RH2s[idx] = (1.f - h * h) * ar[rb][r];
                }
        }
        __syncthreads();
        // ---- output layer and its tangent:  Rmu = H2 vW3 + RH2 W3 + vb3   (wave rb: row block rb)
        if (w < RB) {
            f32x4 am = splat4(b3s[i16]), ar = splat4(vb3s[i16]);
#pragma unroll
            for (int s = 0; s < WIDE_NS; ++s) {
                const int k = wide_kmap<H>(s, kk);
                const float h = H2s[(16 * w + i16) * HS + k], rh = RH2s[(16 * w + i16) * HS + k];
                const float b = W3s[k * W3S + i16], vb = vW3s[k * W3S + i16];
                am = mfma16(h, b, am);
                ar = mfma16(h, vb, ar);
                ar = mfma16(rh, b, ar);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Mss[(16 * w + 4 * kk + r) * MS + i16] = am[r];
                Ms2s[(16 * w + 4 * kk + r) * MS + i16] = ar[r];
            }
        }
        __syncthreads();
        // ---- loss-level R-operator (same arithmetic as k_hvp)
        if (epi) {
            const bool rvalid = erow < nrows;
            const long long n = (long long)base + (rvalid ? erow : 0);
            const float* olsp = a.old_log_std + (a.ls_per_row ? n * A : (long long)task * A);
            const float advn = rvalid ? a.adv[n] : 0.f;
            const float ac0 = a.act[n * A + q0], ac1 = a.act[n * A + q1];
            const float mo0 = a.old_mean[n * A + q0], mo1 = a.old_mean[n * A + q1];
            const float so0 = olsp[q0], so1 = olsp[q1];
            float dlp = 0.f, Rlp = 0.f, kl = 0.f;
            float z0 = 0.f, z1 = 0.f, e0 = 0.f, e1 = 0.f, Rmu0 = 0.f, Rmu1 = 0.f, dklm0 = 0.f, dklm1 = 0.f, dkls0 = 0.f,
                  dkls1 = 0.f, Rs0 = 0.f, Rs1 = 0.f;
            // objective = the mean KL itself (the TRPO constraint, LOSS_KL): R{dKL/dmu}, R{dKL/ds} along v (formulas: k_chain_hvp)
            float kRdm0 = 0.f, kRdm1 = 0.f, kRds0 = 0.f, kRds1 = 0.f;
            if (own0) {
                const float s = lss[q], mu = Mss[erow * MS + q];
                Rmu0 = Ms2s[erow * MS + q];
                Rs0 = vls[q];
                e0 = ess[q];
                z0 = (ac0 - mu) * e0;
                const float zo = (ac0 - mo0) * fast_exp(-so0);
                dlp += (so0 - s) - 0.5f * (z0 * z0 - zo * zo);
                Rlp += z0 * e0 * Rmu0 + (z0 * z0 - 1.f) * Rs0;
                const float sn2 = sn2s[q], num = (mo0 - mu) * (mo0 - mu) + fast_exp(2.f * so0) - sn2, den = 2.f * sn2 + 1e-8f;
                const float rden = fast_rcp(den);
                kl += num * rden + s - so0;
                dklm0 = -2.f * (mo0 - mu) * rden * invN;
                dkls0 = ((-2.f * sn2 * den - 4.f * num * sn2) * (rden * rden) + 1.f) * invN;
                const float D = mo0 - mu, Pk = sn2 * (den + 2.f * num);
                const float RP = 2.f * sn2 * Rs0 * (den + 2.f * num) - 4.f * sn2 * D * Rmu0;
                kRdm0 = (2.f * Rmu0 * rden + 8.f * D * sn2 * Rs0 * (rden * rden)) * invN;
                kRds0 = (-2.f * RP + 16.f * Pk * sn2 * Rs0 * rden) * (rden * rden) * invN;
            }
            if (own1) {
                const float s = lss[q + 4], mu = Mss[erow * MS + q + 4];
                Rmu1 = Ms2s[erow * MS + q + 4];
                Rs1 = vls[q + 4];
                e1 = ess[q + 4];
                z1 = (ac1 - mu) * e1;
                const float zo = (ac1 - mo1) * fast_exp(-so1);
                dlp += (so1 - s) - 0.5f * (z1 * z1 - zo * zo);
                Rlp += z1 * e1 * Rmu1 + (z1 * z1 - 1.f) * Rs1;
                const float sn2 = sn2s[q + 4], num = (mo1 - mu) * (mo1 - mu) + fast_exp(2.f * so1) - sn2, den = 2.f * sn2 + 1e-8f;
                const float rden = fast_rcp(den);
                kl += num * rden + s - so1;
                dklm1 = -2.f * (mo1 - mu) * rden * invN;
                dkls1 = ((-2.f * sn2 * den - 4.f * num * sn2) * (rden * rden) + 1.f) * invN;
                const float D = mo1 - mu, Pk = sn2 * (den + 2.f * num);
                const float RP = 2.f * sn2 * Rs1 * (den + 2.f * num) - 4.f * sn2 * D * Rmu1;
                kRdm1 = (2.f * Rmu1 * rden + 8.f * D * sn2 * Rs1 * (rden * rden)) * invN;
                kRds1 = (-2.f * RP + 16.f * Pk * sn2 * Rs1 * rden) * (rden * rden) * invN;
            }
            dlp += shfl_xor_f32(dlp, 1);  dlp += shfl_xor_f32(dlp, 2);
            Rlp += shfl_xor_f32(Rlp, 1);  Rlp += shfl_xor_f32(Rlp, 2);
            kl += shfl_xor_f32(kl, 1);  kl += shfl_xor_f32(kl, 2);
            float c = 0.f, Rc = 0.f, km = 0.f;
            if (rvalid) {
                km = 1.f;
                if (a.loss_kind == LOSS_RATIO) {
                    c = -advn * expf(dlp) * invN;
                    Rc = c * Rlp;
                } else {
                    c = -advn * invN;
                }
                if (q == 0) klsum += kl * invN;
            }
            const bool klobj = a.loss_kind == LOSS_KL;      // the outputs are MINUS the tangent of the gradient (see the header)
            if (own0) {
                const float Rz = -Rmu0 * e0 - z0 * Rs0;
                const float Rd = Rc * z0 * e0 + c * (Rz * e0 - z0 * e0 * Rs0);
                const float Rds = Rc * (z0 * z0 - 1.f) + 2.f * c * z0 * Rz;
                const float d = klobj ? km * dklm0 : c * z0 * e0;
                const float qm = klobj ? -km * kRdm0 : km * (-Rd + klw * dklm0);
                Mss[erow * MS + q] = d;
                Ms2s[erow * MS + q] = qm;
                outs0 += klobj ? -km * kRds0 : km * (-Rds + klw * dkls0);
                outb30 += qm;
            }
            if (own1) {
                const float Rz = -Rmu1 * e1 - z1 * Rs1;
                const float Rd = Rc * z1 * e1 + c * (Rz * e1 - z1 * e1 * Rs1);
                const float Rds = Rc * (z1 * z1 - 1.f) + 2.f * c * z1 * Rz;
                const float d = klobj ? km * dklm1 : c * z1 * e1;
                const float qm = klobj ? -km * kRdm1 : km * (-Rd + klw * dklm1);
                Mss[erow * MS + q + 4] = d;
                Ms2s[erow * MS + q + 4] = qm;
                outs1 += klobj ? -km * kRds1 : km * (-Rds + klw * dkls1);
                outb31 += qm;
            }
        }
        __syncthreads();
        // ---- out_W3 rows 16w.. += -RH2^T dmu + H2^T qmu ; dZ2 over H2, qZ2 over RH2 (own columns)
        {
#pragma unroll
            for (int s = 0; s < 4 * RB; ++s) {
                const int row = wide_rowmap<RB>(s, kk);
                aw3 = mfma16(RH2s[row * HS + col], -Mss[row * MS + i16], aw3);
                aw3 = mfma16(H2s[row * HS + col], Ms2s[row * MS + i16], aw3);
            }
            f32x4 ad[RB], aq[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) ad[rb] = aq[rb] = zero4();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float b = W3Ts[(4 * s + kk) * HS + col], vb = vW3Ts[(4 * s + kk) * HS + col];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const float dm = Mss[(16 * rb + i16) * MS + 4 * s + kk], qm = Ms2s[(16 * rb + i16) * MS + 4 * s + kk];
                    ad[rb] = mfma16(dm, b, ad[rb]);
                    aq[rb] = mfma16(qm, b, aq[rb]);
                    aq[rb] = mfma16(dm, -vb, aq[rb]);
                }
            }
            wave_sync();
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = (16 * rb + 4 * kk + r) * HS + col;
                    const float h = H2s[idx], rh = RH2s[idx];
                    const float d1 = 1.f - h * h;
                    const float qz = aq[rb][r] * d1 + 2.f * ad[rb][r] * h * rh;
                    H2s[idx] = ad[rb][r] * d1;
                    RH2s[idx] = qz;
                    ob2acc += qz;
                }
        }
        // ---- out_W2 columns 16w.. += -RH1^T dZ2 + H1^T qZ2 ;  qH1 for the same columns
        f32x4 ad1[RB], aq1[RB];
        {
            float wr[WIDE_NS], vr[WIDE_NS];
            sched_fence();
            wide_load_rows<H>(wr, th + oW2, col, kk, 1.f);
            wide_load_rows<H>(vr, v + oW2, col, kk, -1.f);
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 4 * RB; ++s) {
                const int row = wide_rowmap<RB>(s, kk);
                const float dz = -H2s[row * HS + col], qz = RH2s[row * HS + col];
#pragma unroll
                for (int i = 0; i < WIDE_NC; ++i) {
                    aw2[i] = mfma16(RH1s[row * HS + 16 * i + i16], dz, aw2[i]);
                    aw2[i] = mfma16(H1s[row * HS + 16 * i + i16], qz, aw2[i]);
                }
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) ad1[rb] = aq1[rb] = zero4();
            wide_gemm_reg<RB, H>(ad1, H2s, HS, wr, i16, kk);
            wide_gemm_reg<RB, H>(aq1, RH2s, HS, wr, i16, kk);
            wide_gemm_reg<RB, H>(aq1, H2s, HS, vr, i16, kk);      // vr holds -vW2
        }
        __syncthreads();   // every wave has read all of H1 / RH1 before the in-place qZ1
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = (16 * rb + 4 * kk + r) * HS + col;
                const float h = H1s[idx], rh = RH1s[idx];
                const float qz = aq1[rb][r] * (1.f - h * h) + 2.f * ad1[rb][r] * h * rh;
                H1s[idx] = qz;
                ob1acc += qz;
            }
        wave_sync();
        // ---- out_W1 columns 16w.. += X^T qZ1
#pragma unroll
        for (int s = 0; s < 4 * RB; ++s) {
            const int row = wide_rowmap<RB>(s, kk);
            const float b = H1s[row * HS + col];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) aw1[ob] = mfma16(Xs[row * XS + 16 * ob + i16], b, aw1[ob]);
        }
    }

    const int tidt = tid + opaque_zero();          // (the tail's lane indices too: recomputed, not reloaded from a spill)
    const int i16 = tidt & 15, kk = (tidt & 63) >> 4, col = 16 * (tidt >> 6) + i16;
    float* P = a.partials + (long long)wi * a.partial_stride;
#pragma unroll
    for (int m = 4; m <= 32; m <<= 1) {
        outs0 += shfl_xor_f32(outs0, m);  outs1 += shfl_xor_f32(outs1, m);  outb30 += shfl_xor_f32(outb30, m);
        outb31 += shfl_xor_f32(outb31, m);  klsum += shfl_xor_f32(klsum, m);
    }
    __syncthreads();
    if (lane < 4 && w < 2) {       // waves 0..1 ran the epilogue (4 lanes x 32 rows); lane == q
        float* rw = red + 16 * w;
        rw[lane] = outs0;  rw[4 + lane] = outs1;  rw[8 + lane] = outb30;  rw[12 + lane] = outb31;
    }
    float* sc = Xs;
    if (lane == 0 && w < 2) sc[w] = klsum;
    __syncthreads();
    if (tid == 0) {
        P[NP] = 0.f;
        P[NP + 1] = sc[0] + sc[1];
    }
    if (tid < 16) {
        const int j = tid & 3, which = tid >> 2;
        const float t = red[4 * which + j] + red[16 + 4 * which + j];
        const int aidx = j + ((which & 1) ? 4 : 0);
        if (aidx < A) {
            if (which < 2) P[oS + aidx] = t * lmask[aidx];
            else P[ob3 + aidx] = t;
        }
    }
    ob1acc += shfl_xor_f32(ob1acc, 16);  ob1acc += shfl_xor_f32(ob1acc, 32);
    ob2acc += shfl_xor_f32(ob2acc, 16);  ob2acc += shfl_xor_f32(ob2acc, 32);
    if (kk == 0) {
        P[ob1 + col] = ob1acc;
        P[ob2 + col] = ob2acc;
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * ob + 4 * kk + r;
            if (row < O) P[row * H + col] = aw1[ob][r];
        }
#pragma unroll
    for (int i = 0; i < WIDE_NC; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[oW2 + (16 * i + 4 * kk + r) * H + col] = aw2[i][r];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (i16 < A) P[oW3 + (16 * w + 4 * kk + r) * A + i16] = aw3[r];
}
