// promp_kernels_pass.h -- objective + mean KL (+ gradient) of one task slab for hidden widths <= 64 (reference rows a8-a11):
//
//   k_fwd_bwd : objective (ratio / PPO-clip / log-lik / KL), mean KL and gradient on one slab            (K8-K11)
//
// Arithmetic follows oracle/promp.py (which restates meta_algos/pro_mp.py:59-155, meta_algos/base.py:192-215,
// policies/networks/mlp.py:65-119, policies/distributions/diagonal_gaussian.py:16-109 of the reference).
//
// Work decomposition.  A workgroup of 8 waves (two per SIMD) shares its task's parameters in LDS; each wave walks its own
// 16-row tiles through the whole chain with private LDS tiles (no workgroup barrier in the tile loop: the two waves of a
// SIMD drift apart and cover each other's LDS / dependency stalls).  Every GEMM runs on the matrix cores in exact FP32
// (v_mfma_f32_16x16x4_f32), operands read from LDS: activations as [row][unit] with an odd stride, parameters as
// [in][out(+1)].  Weight-gradient tiles stay in registers across all tiles of the wave; the waves' tiles are added in a
// fixed order and written once, as one partial per (workgroup, task); k_reduce_task adds the partials of a task in
// fixed order (bitwise reproducible).
//
// The R-operator pass (k_chain_hvp, promp_kernels_chain.h) uses a different, register-chained design: measured on the
// MI355X it is the faster one where the per-tile state is large (one wave per SIMD, 512 registers), while this
// two-waves-per-SIMD kernel is the faster one for the plain forward / backward pass.
#pragma once
#include "promp_device.h"
#include "promp_kernels_chain.h"

#define PROMP_W3S 17         // row stride of the zero-padded [H2][16] output kernel in LDS
#define PROMP_MS 17          // row stride of the [64][16] mean / d-mean staging tiles

// k_fwd_bwd's work item: the 8 waves of a workgroup are shared out at WAVE granularity, so a workgroup may serve two
// tasks (waves [0, nw0) segment 0, waves [nw0, 8) segment 1).  With 2048 wave slots and 250 tiles per task (config 3)
// every wave walks at most 5 tiles; at workgroup granularity (6 or 7 workgroups per task) 60 % of the tasks had waves
// with 6.  Wave i of a task takes the tiles i, i + wstride, ... of the task's 16-row tiles.
struct PassWork {
    int task[2];      // task of each segment (segment 1 unused when nw0 == 8)
    int row0[2];      // first row of the task
    int nrows[2];     // rows of the task
    int ntiles[2];    // 16-row tiles of the task
    int wave0[2];     // index, among the task's waves, of the segment's first wave
    int wstride[2];   // waves the task has in total
    int slot[2];      // partial-sum row the segment writes
    int nw0, pad;
};

// ---------------------------------------------------------------------------------------------
// k_fwd_bwd -- wave-private pipelines.
//
// The workgroup (4 waves, TWO workgroups resident per CU => 2 waves per SIMD) shares one task's parameters
// in LDS; every wave walks its own 16-row tiles through the whole forward/backward chain with its own LDS
// buffers, so there is NO workgroup barrier inside the tile loop: waves drift apart and one wave's VALU /
// LDS segments overlap the MFMA segments of the other wave on the same SIMD.  All GEMMs are
// v_mfma_f32_16x16x4_f32 (exact FP32, same FLOP rate as 32x32x2) with up to 16 independent accumulators per
// GEMM, which also covers the 40-cycle dependent-accumulator latency.  The cotangent tiles dZ2 / dZ1
// overwrite H2 / H1 in place.  Weight-gradient tiles live in registers across all tiles of the wave; the
// four waves' tiles are added in a fixed order through LDS at the end (bitwise reproducible).
// ---------------------------------------------------------------------------------------------
#define PROMP_WROWS 16
#define PROMP_XS 33

struct LdsWave {
    int w1, b1, w2, b2, w3, w3t, b3, ls, lmask, es, sn2;
    int wave0, wave_stride, x, h1, h2, ms;   // per-wave region: offsets of the private buffers inside it
    int total, HS, WS, Opad4, XS, copy_stride;
};

PROMP_HD LdsWave make_layout_wave(int O, int H1, int H2, int nwaves, int NP) {
    LdsWave L;
    int o = 0;
#define PROMP_TAKE(field, n) \
    L.field = o;             \
    o += ((n) + 3) & ~3
    L.Opad4 = (O + 3) & ~3;
    L.WS = H2 + 1;
    PROMP_TAKE(w1, L.Opad4 * H1);
    PROMP_TAKE(b1, H1);
    PROMP_TAKE(w2, H1 * L.WS);
    PROMP_TAKE(b2, H2);
    PROMP_TAKE(w3, H2 * PROMP_W3S);
    PROMP_TAKE(w3t, 8 * H2);
    PROMP_TAKE(b3, 16);
    PROMP_TAKE(ls, 16);
    PROMP_TAKE(lmask, 16);
    PROMP_TAKE(es, 16);
    PROMP_TAKE(sn2, 16);
    L.copy_stride = o;          // one task's parameter block; a second copy follows for the other segment's task
    o *= 2;
    L.HS = (H1 > H2 ? H1 : H2) + 1;
    L.wave0 = o;
    int q = 0;
    // X tile [16][XS]: the hidden_0 gradient reads it transposed with observation indices up to 31; indices >= O land
    // in the following row / buffer (finite values) and only produce gradient rows >= O, which are never written out
    L.XS = L.Opad4 + 1;
    L.x = q;  q += (PROMP_WROWS * L.XS + 3) & ~3;
    L.h1 = q; q += (PROMP_WROWS * L.HS + 3) & ~3;
    L.h2 = q; q += (PROMP_WROWS * L.HS + 3) & ~3;
    L.ms = q; q += (PROMP_WROWS * PROMP_MS + 3) & ~3;
    L.wave_stride = q;
    o += nwaves * q;
    {   // end-of-kernel: one slab per wave from offset 0 (aliases everything): the hidden_1 kernel with rows padded to
        // H2 + 4, then the hidden_0 kernel (32 rows padded to H1 + 4) followed by everything else
        const int nw2 = H1 * (H2 + 4), nr2 = 32 * (H1 + 4) + (NP + 2 - H1 * H2);
        const int need = nwaves * (nw2 > nr2 ? nw2 : nr2);
        if (o < need) o = need;
    }
#undef PROMP_TAKE
    L.total = o;
    return L;
}

// BWD = false: objective and mean KL only (compute_stats / line-search evaluations): the tile loop stops after the
// distribution epilogue and the partial carries just the two scalars.
// KS1 > 0: the observation width is known at compile time (ceil(O / 4) == KS1 k-steps in the first layer): the k-loop
// unrolls and its accumulators stay in place (the runtime loop pays a register copy per accumulator element and step).
// STORE: the hidden activations, the means and the hidden_0 cotangent (before its tanh derivative) of every tile also go to
// the step's primal cache (chain_cache_row in promp_kernels_chain.h) for the R-operator pass that follows at the same
// parameters: 52 four-byte stores per lane and tile.
template <int NB1, int NB2, int NW, bool BWD, int KS1 = 0, bool STORE = false>
__global__ void __launch_bounds__(64 * NW, NW / 4) k_fwd_bwd(PassArgs a) {
    constexpr int NT = 64 * NW;
    constexpr int H1 = 32 * NB1, H2 = 32 * NB2, NC1 = H1 / 16, NC2 = H2 / 16, MS = PROMP_MS, W3S = PROMP_W3S;
    constexpr int Q1 = H1 / 4, Q2 = H2 / 4;   // k-slice of lane group kk in the K = H GEMMs: {kk*Q .. kk*Q + Q-1}
    PROMP_SMEM_DECL;
    float* sm = (float*)PROMP_SMEM_PTR;
    const int tid = threadIdx.x, lane = tid & 63, w_ = tid >> 6;
    const int i16 = lane & 15, kk = lane >> 4;
    CH_WGSTAMP(0);
    CH_STAMP(0);
    const PassWork pw = a.pwork[blockIdx.x];
    const int w = wave_uniform(w_);
    const int seg = (w < pw.nw0) ? 0 : 1;      // which of the workgroup's (at most two) tasks this wave serves
    const int task = pw.task[seg];
    const int O = a.O, A = a.A;
    const int ob1 = O * H1, oW2 = ob1 + H1, ob2 = oW2 + H1 * H2, oW3 = ob2 + H2, ob3 = oW3 + H2 * A, oS = ob3 + A,
              NP = oS + A;
    // the layout is sized for obs_dim 32 whatever O is: every LDS offset is then a compile-time constant and folds into
    // the ds_read / ds_write immediates instead of costing address arithmetic in the tile loop
    const LdsWave L = make_layout_wave(32, H1, H2, NW, 0);
    const int HS = L.HS, WS = L.WS, Opad4 = (O + 3) & ~3, XS = L.XS;
    float* wreg = sm + L.wave0 + w * L.wave_stride;
    float *Xw = wreg + L.x, *H1w = wreg + L.h1, *H2w = wreg + L.h2, *Msw = wreg + L.ms;
    const float invN = 1.0f / (float)pw.nrows[seg];

    // ---- stage the parameters of the workgroup's task(s) (shared by the waves of a segment): every global load is
    //      issued before the first LDS store, so the staging costs one L2 round trip instead of one per loop iteration ----
    for (int sg = 0; sg < (pw.nw0 < NW ? 2 : 1); ++sg) {
        float* cp = sm + sg * L.copy_stride;
        float *W1s = cp + L.w1, *b1s = cp + L.b1, *W2s = cp + L.w2, *b2s = cp + L.b2, *W3s = cp + L.w3, *W3Ts = cp + L.w3t,
              *b3s = cp + L.b3, *lss = cp + L.ls, *lmask = cp + L.lmask, *ess = cp + L.es, *sn2s = cp + L.sn2;
        const float* th = a.theta + (long long)pw.task[sg] * a.theta_task_stride;
        constexpr int N1 = (32 * H1 + NT - 1) / NT, N2 = H1 * H2 / NT, N3 = (H2 * 16 + NT - 1) / NT, N3T = (8 * H2 + NT - 1) / NT;
        float r1[N1], r2[N2], r3[N3], r3t[N3T];
#pragma unroll
        for (int i = 0; i < N1; ++i) {
            const int e = tid + i * NT;
            r1[i] = (e < O * H1) ? PROMP_TANH_PRESCALE * th[e] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < N2; ++i) r2[i] = PROMP_TANH_PRESCALE * th[oW2 + tid + i * NT];
#pragma unroll
        for (int i = 0; i < N3; ++i) {
            const int e = tid + i * NT, k = e >> 4, j = e & 15;
            r3[i] = (e < H2 * 16 && j < A) ? th[oW3 + k * A + j] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < N3T; ++i) {
            const int e = tid + i * NT, aa = e / H2, k = e - aa * H2;
            r3t[i] = (e < 8 * H2 && aa < A) ? th[oW3 + k * A + aa] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < N1; ++i) {
            const int e = tid + i * NT;
            if (e < Opad4 * H1) W1s[e] = r1[i];
        }
#pragma unroll
        for (int i = 0; i < N2; ++i) {
            const int e = tid + i * NT, k = e / H2, j = e - k * H2;
            W2s[k * WS + j] = r2[i];
        }
#pragma unroll
        for (int i = 0; i < N3; ++i) {
            const int e = tid + i * NT, k = e >> 4, j = e & 15;
            if (e < H2 * 16) W3s[k * W3S + j] = r3[i];
        }
#pragma unroll
        for (int i = 0; i < N3T; ++i) {
            const int e = tid + i * NT;
            if (e < 8 * H2) W3Ts[e] = r3t[i];
        }
        if (tid < H1) b1s[tid] = PROMP_TANH_PRESCALE * th[ob1 + tid];
        if (tid < H2) b2s[tid] = PROMP_TANH_PRESCALE * th[ob2 + tid];
        if (tid < 16) {
            b3s[tid] = (tid < A) ? th[ob3 + tid] : 0.f;
            const float sr = (tid < A) ? th[oS + tid] : 0.f;
            const bool clipped = a.clip_log_std && (sr < a.min_log_std);   // tf.maximum: gradient iff var >= min
            const float s = clipped ? a.min_log_std : sr;
            lss[tid] = s;
            lmask[tid] = clipped ? 0.f : 1.f;
            ess[tid] = expf(-s);
            sn2s[tid] = expf(2.f * s);
        }
    }
    // this wave's view of its task's parameter block
    float* const cp = sm + seg * L.copy_stride;
    float *W1s = cp + L.w1, *b1s = cp + L.b1, *W2s = cp + L.w2, *b2s = cp + L.b2, *W3s = cp + L.w3, *W3Ts = cp + L.w3t,
          *b3s = cp + L.b3, *lss = cp + L.ls, *lmask = cp + L.lmask, *ess = cp + L.es, *sn2s = cp + L.sn2;
    for (int e = lane; e < L.wave_stride; e += 64) wreg[e] = 0.f;   // pad columns stay zero; over-read cells finite
    __syncthreads();
    CH_STAMP(1);

    // ---- persistent accumulators of this wave ----
    f32x4 aw2[NC1][NC2], aw1[2][NC1], aw3[NC2][1];
#pragma unroll
    for (int i = 0; i < NC1; ++i)
#pragma unroll
        for (int j = 0; j < NC2; ++j) aw2[i][j] = zero4();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NC1; ++j) aw1[i][j] = zero4();
#pragma unroll
    for (int j = 0; j < NC2; ++j) aw3[j][0] = zero4();
    float gb1[NC1], gb2[NC2];
#pragma unroll
    for (int j = 0; j < NC1; ++j) gb1[j] = 0.f;
#pragma unroll
    for (int j = 0; j < NC2; ++j) gb2[j] = 0.f;
    float loss = 0.f, klsum = 0.f, gs0 = 0.f, gs1 = 0.f, gb30 = 0.f, gb31 = 0.f;

    // epilogue role: 4 lanes per row, actions {q, q+4}
    const int erow = lane >> 2, q = lane & 3;
    const bool own0 = q < A, own1 = (q + 4) < A;
    const int q0 = own0 ? q : 0, q1 = own1 ? q + 4 : 0;   // clamped action indices for unconditional loads
    // this lane's share of a [16][O] tile lands at row e / O, column e % O of the padded LDS tile (e = lane + 64 u);
    // the quotient comes from a float reciprocal (exact for these small integers) instead of 8 live offset registers
    const float rO = 1.0f / (float)O;
    // this wave's tiles of its task: wi, wi + wstride, ...
    const int wi = pw.wave0[seg] + (seg ? w - pw.nw0 : w), wstride = pw.wstride[seg], ntiles = pw.ntiles[seg];
    const int trow0 = pw.row0[seg], tnrows = pw.nrows[seg];
    float xr[8];
    {
        const int nr = (tnrows - PROMP_WROWS * wi) < PROMP_WROWS ? (tnrows - PROMP_WROWS * wi) : PROMP_WROWS;
        const int lim = (wi < ntiles) ? nr * O : 0;
        const float* src = a.obs + (long long)(trow0 + (wi < ntiles ? PROMP_WROWS * wi : 0)) * O;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = lane + 64 * u;
            const float x = src[e < lim ? e : 0];      // always a valid address: no exec-masked branch per load
            xr[u] = (e < lim) ? x : 0.f;
        }
    }

    const unsigned hoff = 64 * kk + i16;     // STORE: this lane's offset inside a cache block
    int tix = 0;
#ifndef PROMP_NO_TILE_PRIO
    // The two waves of a SIMD share its issue slots oldest-first, so the older wave runs ahead and the younger one is
    // left to finish alone (one wave per SIMD pays twice as many issue cycles per VALU / LDS instruction): the wave with
    // more tiles left gets the higher priority, which keeps the pair within a tile of each other up to the end.
    const int my_tiles = (wi < ntiles) ? (ntiles - wi + wstride - 1) / wstride : 0;
#endif
    for (int t = wi; t < ntiles; t += wstride, ++tix) {
#ifndef PROMP_NO_TILE_PRIO
        wave_priority(my_tiles - 1 - tix);
#endif
        CH_TSTAMP(0);
        const int base = trow0 + PROMP_WROWS * t;
        const int nrows = (tnrows - PROMP_WROWS * t) < PROMP_WROWS ? (tnrows - PROMP_WROWS * t) : PROMP_WROWS;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = lane + 64 * u;
            const int r = (int)(((float)e + 0.5f) * rO);
            if (e < PROMP_WROWS * O) Xw[r * XS + (e - r * O)] = xr[u];
        }
        {
            const int tn = t + wstride;
            const int nn = (tnrows - PROMP_WROWS * tn) < PROMP_WROWS ? (tnrows - PROMP_WROWS * tn) : PROMP_WROWS;
            const int lim = (tn < ntiles) ? nn * O : 0;
            const float* src = a.obs + (long long)(trow0 + (tn < ntiles ? PROMP_WROWS * tn : 0)) * O;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = lane + 64 * u;
                const float x = src[e < lim ? e : 0];
                xr[u] = (e < lim) ? x : 0.f;
            }
        }
        // STORE: this lane's cells of the tile's cache block are sample 4 kk (+ r), unit i16 (+ 16 j) / action i16
        // (a wave-uniform block address in scalar registers + one constant 32-bit offset per lane)
        float* const hcb = STORE ? a.hcache + ((long long)base + 16 * task) * chain_cache_row(H1, H2) : nullptr;
        const bool rvalid = erow < nrows;
        const long long n = (long long)base + (rvalid ? erow : 0);
        const float* olsp = a.old_log_std + (a.ls_per_row ? n * A : (long long)task * A);
        // n is a valid row even for padding lanes; q0/q1 are valid action indices even for lanes that own none:
        // all loads are unconditional, the selects below discard what is not owned
        const float advn = rvalid ? a.adv[n] : 0.f;
        const float ac0 = a.act[n * A + q0], ac1 = a.act[n * A + q1];
        const float mo0 = a.old_mean[n * A + q0], mo1 = a.old_mean[n * A + q1];
        const float so0 = olsp[q0], so1 = olsp[q1];
        wave_sync();
        // ---- layer 1: H1 = tanh(X W1 + b1)
        {
            f32x4 acc[1][NC1];
#pragma unroll
            for (int j = 0; j < NC1; ++j) acc[0][j] = splat4(b1s[16 * j + i16]);   // bias rides in the accumulator
            outer16<1, NC1>(acc, Xw + i16 * XS + kk, 4, 0, W1s + kk * H1 + i16, 4 * H1, 16, KS1 > 0 ? KS1 : Opad4 / 4, 1.f);
#pragma unroll
            for (int j = 0; j < NC1; ++j)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2 h = tanh2_prescaled(acc[0][j][r], acc[0][j][r + 1]);
                    H1w[(4 * kk + r) * HS + 16 * j + i16] = h[0];
                    H1w[(4 * kk + r + 1) * HS + 16 * j + i16] = h[1];
                    if (STORE) {
                        hcb[hoff + 256 * j + 16 * r] = h[0];
                        hcb[hoff + 256 * j + 16 * (r + 1)] = h[1];
                    }
                }
        }
        wave_sync();
        CH_TSTAMP(1);
        // ---- layer 2
        {
            f32x4 acc[1][NC2];
#pragma unroll
            for (int j = 0; j < NC2; ++j) acc[0][j] = splat4(b2s[16 * j + i16]);
            outer16<1, NC2>(acc, H1w + i16 * HS + kk * Q1, 1, 0, W2s + kk * Q1 * WS + i16, WS, 16, Q1, 1.f);
#pragma unroll
            for (int j = 0; j < NC2; ++j)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2 h = tanh2_prescaled(acc[0][j][r], acc[0][j][r + 1]);
                    H2w[(4 * kk + r) * HS + 16 * j + i16] = h[0];
                    H2w[(4 * kk + r + 1) * HS + 16 * j + i16] = h[1];
                    if (STORE) {
                        hcb[hoff + 256 * (NC1 + j) + 16 * r] = h[0];
                        hcb[hoff + 256 * (NC1 + j) + 16 * (r + 1)] = h[1];
                    }
                }
        }
        wave_sync();
        CH_TSTAMP(2);
        // ---- output layer (16 padded columns)
        {
            f32x4 acc[1][1];
            acc[0][0] = splat4(b3s[i16]);
            outer16<1, 1>(acc, H2w + i16 * HS + kk * Q2, 1, 0, W3s + kk * Q2 * W3S + i16, W3S, 0, Q2, 1.f);
#pragma unroll
            for (int r = 0; r < 4; ++r) Msw[(4 * kk + r) * MS + i16] = acc[0][0][r];
            if (STORE && i16 < 8) {
#pragma unroll
                for (int r = 0; r < 4; ++r) hcb[hoff + 256 * (NC1 + NC2) - 32 * kk - (i16 & 8) + 8 * r] = acc[0][0][r];
            }
        }
        wave_sync();
        CH_TSTAMP(3);
        // ---- distribution + objective epilogue
        {
            float dlp = 0.f, sumz2 = 0.f, sums = 0.f, kl = 0.f;
            float z0 = 0.f, z1 = 0.f, e0 = 0.f, e1 = 0.f, dklm0 = 0.f, dklm1 = 0.f, dkls0 = 0.f, dkls1 = 0.f;
            if (own0) {
                const float s = lss[q], mu = Msw[erow * MS + q];
                e0 = ess[q];
                z0 = (ac0 - mu) * e0;
                const float zo = (ac0 - mo0) * fast_exp(-so0);
                dlp += (so0 - s) - 0.5f * (z0 * z0 - zo * zo);
                sumz2 += z0 * z0;
                sums += s;
                const float sn2 = sn2s[q], num = (mo0 - mu) * (mo0 - mu) + fast_exp(2.f * so0) - sn2, den = 2.f * sn2 + 1e-8f;
                const float rden = fast_rcp(den);     // one v_rcp_f32 (1 ulp) serves the KL and both of its cotangents
                kl += num * rden + s - so0;
                dklm0 = -2.f * (mo0 - mu) * rden;
                dkls0 = (-2.f * sn2 * den - 4.f * num * sn2) * (rden * rden) + 1.f;
            }
            if (own1) {
                const float s = lss[q + 4], mu = Msw[erow * MS + q + 4];
                e1 = ess[q + 4];
                z1 = (ac1 - mu) * e1;
                const float zo = (ac1 - mo1) * fast_exp(-so1);
                dlp += (so1 - s) - 0.5f * (z1 * z1 - zo * zo);
                sumz2 += z1 * z1;
                sums += s;
                const float sn2 = sn2s[q + 4], num = (mo1 - mu) * (mo1 - mu) + fast_exp(2.f * so1) - sn2, den = 2.f * sn2 + 1e-8f;
                const float rden = fast_rcp(den);     // one v_rcp_f32 (1 ulp) serves the KL and both of its cotangents
                kl += num * rden + s - so1;
                dklm1 = -2.f * (mo1 - mu) * rden;
                dkls1 = (-2.f * sn2 * den - 4.f * num * sn2) * (rden * rden) + 1.f;
            }
            dlp += shfl_xor_f32(dlp, 1);  dlp += shfl_xor_f32(dlp, 2);
            sumz2 += shfl_xor_f32(sumz2, 1);  sumz2 += shfl_xor_f32(sumz2, 2);
            sums += shfl_xor_f32(sums, 1);  sums += shfl_xor_f32(sums, 2);
            kl += shfl_xor_f32(kl, 1);  kl += shfl_xor_f32(kl, 2);
            float c = 0.f, ck = 0.f;   // d loss / d logpi, and the weight of the KL cotangents (LOSS_KL only)
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
                Msw[erow * MS + q] = d;
                gs0 += c * (z0 * z0 - 1.f) + ck * dkls0;
                gb30 += d;
            }
            if (own1) {
                const float d = c * z1 * e1 + ck * dklm1;
                Msw[erow * MS + q + 4] = d;
                gs1 += c * (z1 * z1 - 1.f) + ck * dkls1;
                gb31 += d;
            }
            // columns >= A of Msw already hold exact zeros (zero-padded W3s / b3s)
        }
        wave_sync();
        CH_TSTAMP(4);
        if (BWD) {
        // ---- output-kernel gradient (+=); dZ2 = (dmu W3^T) * (1 - H2^2) in place over H2
        outer16<NC2, 1>(aw3, H2w + kk * HS + i16, 4 * HS, 16, Msw + kk * MS + i16, 4 * MS, 0, PROMP_WROWS / 4, 1.f);
        sched_fence();
        CH_TSTAMP(5);
        {
            f32x4 acc[1][NC2];
#pragma unroll
            for (int j = 0; j < NC2; ++j) acc[0][j] = zero4();
            outer16<1, NC2>(acc, Msw + i16 * MS + kk, 4, 0, W3Ts + kk * H2 + i16, 4 * H2, 16, 2, 1.f);
#pragma unroll
            for (int j = 0; j < NC2; ++j) {
                float cs = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const int idx = (4 * kk + r) * HS + 16 * j + i16;
                    const f32x2 ns = neg_dtanh2(H2w[idx], H2w[idx + HS]);
                    const float d0 = acc[0][j][r] * -ns[0], d1 = acc[0][j][r + 1] * -ns[1];
                    H2w[idx] = d0;
                    H2w[idx + HS] = d1;
                    cs += d0;
                    cs += d1;
                }
                gb2[j] += cs;
            }
        }
        wave_sync();
        CH_TSTAMP(6);
        // ---- hidden_1 kernel gradient (+=); dZ1 = (dZ2 W2^T) * (1 - H1^2) in place over H1
        outer16<NC1, NC2>(aw2, H1w + kk * HS + i16, 4 * HS, 16, H2w + kk * HS + i16, 4 * HS, 16, PROMP_WROWS / 4, 1.f);
        sched_fence();
        CH_TSTAMP(7);
        {
            f32x4 acc[1][NC1];
#pragma unroll
            for (int j = 0; j < NC1; ++j) acc[0][j] = zero4();
            outer16<1, NC1>(acc, H2w + i16 * HS + kk * Q2, 1, 0, W2s + i16 * WS + kk * Q2, 1, 16 * WS, Q2, 1.f);
#pragma unroll
            for (int j = 0; j < NC1; ++j) {
                float cs = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const int idx = (4 * kk + r) * HS + 16 * j + i16;
                    const f32x2 ns = neg_dtanh2(H1w[idx], H1w[idx + HS]);
                    if (STORE) {
                        hcb[hoff + 256 * (NC1 + NC2) + 128 + 256 * j + 16 * r] = acc[0][j][r];
                        hcb[hoff + 256 * (NC1 + NC2) + 128 + 256 * j + 16 * (r + 1)] = acc[0][j][r + 1];
                    }
                    const float d0 = acc[0][j][r] * -ns[0], d1 = acc[0][j][r + 1] * -ns[1];
                    H1w[idx] = d0;
                    H1w[idx + HS] = d1;
                    cs += d0;
                    cs += d1;
                }
                gb1[j] += cs;
            }
        }
        wave_sync();
        CH_TSTAMP(8);
        // ---- hidden_0 kernel gradient (+=): rows = observation index (two 16-blocks cover O <= 32)
        outer16<2, NC1>(aw1, Xw + kk * XS + i16, 4 * XS, 16, H1w + kk * HS + i16, 4 * HS, 16, PROMP_WROWS / 4, 1.f);
        wave_sync();
        CH_TSTAMP(9);
        }
    }
    CH_STAMP(2);
#ifdef PROMP_DEV_STAMPS
    if (a.dbg != nullptr && blockIdx.x < 4 && lane == 0) { a.dbg[208 + 8 * blockIdx.x + w] = promp_clock(); a.dbg[128 + 8 * blockIdx.x + w] = tix; }
#endif

    if (BWD) {
        // the hidden_1 kernel is staged pre-scaled by PROMP_TANH_PRESCALE (so that the forward pass feeds v_exp_f32 without a
        // multiply); the backward product through it therefore carried that factor into dZ1, i.e. into the hidden_0
        // kernel / bias gradients, and leaves here
        constexpr float inv_c = 1.0f / PROMP_TANH_PRESCALE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NC1; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) aw1[i][j][r] *= inv_c;
#pragma unroll
        for (int j = 0; j < NC1; ++j) gb1[j] *= inv_c;
    }
    const float lmask_reg0 = lmask[lane & 3], lmask_reg1 = lmask[(lane & 3) + 4];
    // ---- add the four waves' results in wave order, then one coalesced partial ----
    // bias sums: lanes with equal i16 hold different row groups -> fold kk
#pragma unroll
    for (int j = 0; j < NC1; ++j) {
        gb1[j] += shfl_xor_f32(gb1[j], 16);
        gb1[j] += shfl_xor_f32(gb1[j], 32);
    }
#pragma unroll
    for (int j = 0; j < NC2; ++j) {
        gb2[j] += shfl_xor_f32(gb2[j], 16);
        gb2[j] += shfl_xor_f32(gb2[j], 32);
    }
    {   // per-action sums over the rows of this wave: lanes with equal q differ in bits 2..5
#pragma unroll
        for (int m = 4; m <= 32; m <<= 1) {
            gs0 += shfl_xor_f32(gs0, m);  gs1 += shfl_xor_f32(gs1, m);  gb30 += shfl_xor_f32(gb30, m);
            gb31 += shfl_xor_f32(gb31, m);  loss += shfl_xor_f32(loss, m);  klsum += shfl_xor_f32(klsum, m);
        }
    }
    const bool two = pw.nw0 < NW;                  // the workgroup served two tasks
    float* P0 = a.partials + (long long)pw.slot[0] * a.partial_stride;
    float* P1 = a.partials + (long long)pw.slot[two ? 1 : 0] * a.partial_stride;
    if (!BWD) {   // only the two scalars per segment leave the workgroup
        float* SC = sm;
        lds_barrier();
        if (lane == 0) {
            SC[2 * w] = loss;
            SC[2 * w + 1] = klsum;
        }
        lds_barrier();
        if (tid < 2 && (tid == 0 || two)) {
            float l = 0.f, k = 0.f;
            const int lo = tid ? pw.nw0 : 0, hi = tid ? NW : pw.nw0;
            for (int ww = lo; ww < hi; ++ww) {
                l += SC[2 * ww];
                k += SC[2 * ww + 1];
            }
            float* Pq = tid ? P1 : P0;
            Pq[NP] = l;
            Pq[NP + 1] = k;
        }
        return;
    }
    // Every wave stores its tiles to its own LDS slab (plain stores, no read-modify-write); then all threads add the slabs
    // in wave order, the waves of segment 0 into the first task's partial and those of segment 1 into the second's.
    // Two payload rounds because NW x [NP] does not fit in LDS: the hidden_1 kernel, then everything else (compacted).
    // Slab rows are padded by 4 floats: the four lane groups of an accumulator tile (rows 4 kk + r) then fall into two
    // bank halves instead of one (a [64]-float row stride puts all four on the same 16 banks).  Every cell of a slab is
    // written by exactly one lane, so nothing is cleared first.
    float* S = sm;                                   // whole LDS allocation is free now
    constexpr int SP2 = H2 + 4, SLAB1 = H1 * SP2;    // round 1: hidden_1 kernel, padded rows
    constexpr int SP1 = H1 + 4, RESTB = 32 * SP1;    // round 2: hidden_0 kernel rows [0, 32) padded, the rest behind them
    const int NW2 = H1 * H2;
    const int NR2 = NP + 2 - NW2;                    // compact index space of round 2
    const int SLAB2 = RESTB + (NR2 - ob1);
    CH_STAMP(4);
    lds_barrier();
    CH_STAMP(5);
    {
        float* mine = S + w * SLAB1;
#pragma unroll
        for (int i = 0; i < NC1; ++i)
#pragma unroll
            for (int j = 0; j < NC2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[(16 * i + 4 * kk + r) * SP2 + 16 * j + i16] = aw2[i][j][r];
    }
    lds_barrier();
    CH_STAMP(6);
#pragma unroll 2
    for (int e = tid; e < NW2; e += NT) {
        const int src = (e / H2) * SP2 + (e % H2);
        float v[NW];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) v[ww] = S[ww * SLAB1 + src];      // all slab reads in flight together
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            if (ww < pw.nw0) t0 += v[ww];
            else t1 += v[ww];
        }
        P0[oW2 + e] = t0;
        if (two) P1[oW2 + e] = t1;
    }
    CH_STAMP(7);
    lds_barrier();        // (LDS ordering only: the partial-row stores in flight are not waited for)
    {
        // round 2: [0, RESTB) hidden_0 kernel rows (padded) | then bias_0 and everything after the hidden_1 kernel, in
        // parameter order; `rest` maps a parameter index >= ob1 (hidden_1 kernel cut out) to its cell
        float* mine = S + w * SLAB2;
        float* rest = mine + RESTB - ob1;            // rest[ob1 + u] = bias_0[u];  rest[p - NW2] for parameters p >= ob2
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NC1; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[(16 * i + 4 * kk + r) * SP1 + 16 * j + i16] = aw1[i][j][r];
#pragma unroll
        for (int j = 0; j < NC2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (i16 < A) rest[oW3 - NW2 + (16 * j + 4 * kk + r) * A + i16] = aw3[j][0][r];
        if (kk == 0) {
#pragma unroll
            for (int j = 0; j < NC1; ++j) rest[ob1 + 16 * j + i16] = gb1[j];
#pragma unroll
            for (int j = 0; j < NC2; ++j) rest[ob2 - NW2 + 16 * j + i16] = gb2[j];
        }
        if (lane < 4) {   // lane == q
            if (lane < A) {
                rest[ob3 - NW2 + lane] = gb30;
                rest[oS - NW2 + lane] = gs0 * lmask_reg0;
            }
            if (lane + 4 < A) {
                rest[ob3 - NW2 + lane + 4] = gb31;
                rest[oS - NW2 + lane + 4] = gs1 * lmask_reg1;
            }
        }
        if (lane == 0) {
            rest[NP - NW2] = loss;
            rest[NP + 1 - NW2] = klsum;
        }
    }
    CH_STAMP(200);
    lds_barrier();
    CH_STAMP(201);
    for (int e = tid; e < NR2; e += NT) {
        const int dst = e < oW2 ? e : e + NW2;
        // compact index e: hidden_0 kernel entries [0, ob1) sit in padded rows, the others behind them
        const int src = e < ob1 ? (e / H1) * SP1 + (e % H1) : RESTB - ob1 + e;
        float v[NW];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) v[ww] = S[ww * SLAB2 + src];
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            if (ww < pw.nw0) t0 += v[ww];
            else t1 += v[ww];
        }
        P0[dst] = t0;
        if (two) P1[dst] = t1;
    }
    CH_STAMP(3);
    CH_WGSTAMP(1);
}
