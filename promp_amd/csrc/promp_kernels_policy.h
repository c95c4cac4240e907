// promp_kernels_policy.h -- per-task Gaussian-MLP policy passes (reference rows a8-a13).
//
//   k_fwd_bwd : objective + mean KL + gradient of one task on one slab            (K8-K11)
//   k_hvp     : out = -H v + kl_weight * grad KL, H = Hessian of the inner objective (K12, K13)
//   k_reduce_*: fixed-order reduction of per-workgroup partials, inner SGD step, task mean (K10, K14)
//   k_adam    : tf.train.AdamOptimizer step                                         (K14)
//
// Arithmetic follows oracle/promp.py (which restates meta_algos/pro_mp.py:59-155,
// meta_algos/base.py:192-215, policies/networks/mlp.py:65-119,
// policies/distributions/diagonal_gaussian.py:16-109 of the reference).
//
// Work decomposition.  A workgroup owns a contiguous row range of ONE task and shares that task's parameters in LDS;
// each of its waves walks its own 16-row tiles through the whole chain with private LDS tiles (no workgroup barrier in
// the tile loop).  Every GEMM runs on the matrix cores in exact FP32 (v_mfma_f32_16x16x4_f32), operands read from LDS:
// activations as [row][unit] with an odd stride, parameters as [in][out(+1)].  Weight-gradient tiles stay in registers
// across all tiles of the wave; the waves' tiles are added in a fixed order and written once, as one partial per
// workgroup; a second tiny kernel adds the partials of a task in fixed order (bitwise reproducible).
#pragma once
#include "promp_device.h"

#define PROMP_W3S 17         // row stride of the zero-padded [H2][16] output kernel in LDS
#define PROMP_MS 17          // row stride of the [64][16] mean / d-mean staging tiles
#define PROMP_PARTIAL_EXTRA 4  // loss, kl, 2 spare

struct WorkItem {
    int task, row_begin, row_end, pad;
};

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

enum { LOSS_RATIO = 0, LOSS_CLIP = 1, LOSS_LOGLIK = 2, LOSS_KL = 3 };   // LOSS_KL: mean KL(old || new) itself (TRPO constraint)

struct PassArgs {
    const float* obs;           // [rows][O]
    const float* act;           // [rows][A]
    const float* adv;           // [rows]
    const float* old_mean;      // [rows][A]
    const float* old_log_std;   // [rows][A] or [tasks][A]
    int ls_per_row;
    const int* task_row_offsets;  // [tasks+1]
    const WorkItem* work;
    const PassWork* pwork;      // k_fwd_bwd only         // [grid]
    const float* theta;           // [Theta] or [tasks][Theta]
    long long theta_task_stride;  // 0 => shared
    const float* vdir;            // hvp: [tasks][Theta]
    float* partials;              // [grid][partial_stride]
    int partial_stride;
    int O, A;
    int loss_kind;
    float clip_eps;
    int clip_log_std;
    float min_log_std;
    float kl_weight;
    unsigned long long* dbg;      // optional cycle stamps of block 0 / lane 0 (tools/phase_timing.py), else NULL
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
// developer tooling: cycle stamps of workgroup 0 / thread 0, kept in LDS and dumped when the kernel ends
// (compiled in only with -DPROMP_DEV_STAMPS: each stamp is a divergent region that splits the scheduler's basic blocks)
#ifdef PROMP_DEV_STAMPS
#define PROMP_STAMP(i) do { if (a.dbg != nullptr && blockIdx.x == 0 && tid == 0) dbgs[(i)] = promp_clock(); } while (0)
#define PROMP_STAMPS_ON 1
#else
#define PROMP_STAMP(i) do { } while (0)
#define PROMP_STAMPS_ON 0
#endif
#define PROMP_WROWS 16
PROMP_DEV f32x4 splat4(float v) {
    f32x4 z;
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = v;
    return z;
}
#define PROMP_XS 33

struct LdsWave {
    int w1, b1, w2, b2, w3, w3t, b3, ls, lmask, es, sn2;
    int wave0, wave_stride, x, h1, h2, ms;   // per-wave region: offsets of the private buffers inside it
    int total, HS, WS, Opad4, dbg, XS, copy_stride;
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
    {   // end-of-kernel: one slab of max(H1*H2, NP+2-H1*H2) floats per wave, from offset 0 (aliases everything)
        const int nw2 = H1 * H2, nr2 = NP + 2 - nw2;
        const int need = nwaves * (nw2 > nr2 ? nw2 : nr2);
        if (o < need) o = need;
    }
    L.dbg = o;
    o += 256;   // 128 cycle stamps (developer tooling)
#undef PROMP_TAKE
    L.total = o;
    return L;
}

// BWD = false: objective and mean KL only (compute_stats / line-search evaluations): the tile loop stops after the
// distribution epilogue and the partial carries just the two scalars.
template <int NB1, int NB2, int NW, bool BWD>
__global__ void __launch_bounds__(64 * NW, NW / 4) k_fwd_bwd(PassArgs a) {
    constexpr int NT = 64 * NW;
    constexpr int H1 = 32 * NB1, H2 = 32 * NB2, NC1 = H1 / 16, NC2 = H2 / 16, MS = PROMP_MS, W3S = PROMP_W3S;
    constexpr int Q1 = H1 / 4, Q2 = H2 / 4;   // k-slice of lane group kk in the K = H GEMMs: {kk*Q .. kk*Q + Q-1}
    PROMP_SMEM_DECL;
    float* sm = (float*)PROMP_SMEM_PTR;
    const int tid = threadIdx.x, lane = tid & 63, w_ = tid >> 6;
    const int i16 = lane & 15, kk = lane >> 4;
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
    unsigned long long* dbgs = (unsigned long long*)(sm + make_layout_wave(32, H1, H2, NW, NP).dbg);   // (after the slabs: needs NP)
    if (PROMP_STAMPS_ON && a.dbg != nullptr && blockIdx.x == 0 && tid < 128) dbgs[tid] = 0;
    float* wreg = sm + L.wave0 + w * L.wave_stride;
    float *Xw = wreg + L.x, *H1w = wreg + L.h1, *H2w = wreg + L.h2, *Msw = wreg + L.ms;
    const float invN = 1.0f / (float)pw.nrows[seg];
    PROMP_STAMP(0);

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
            r1[i] = (e < O * H1) ? th[e] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < N2; ++i) r2[i] = th[oW2 + tid + i * NT];
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
        if (tid < H1) b1s[tid] = th[ob1 + tid];
        if (tid < H2) b2s[tid] = th[ob2 + tid];
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
    PROMP_STAMP(1);

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

    int tix = 0;
    for (int t = wi; t < ntiles; t += wstride, ++tix) {
        PROMP_STAMP(8 + 16 * (tix < 3 ? tix : 3) + 0);
        const int base = trow0 + PROMP_WROWS * t;
        const int nrows = (tnrows - PROMP_WROWS * t) < PROMP_WROWS ? (tnrows - PROMP_WROWS * t) : PROMP_WROWS;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = lane + 64 * u;
            const int r = (int)(((float)e + 0.5f) * rO);
            if (e < PROMP_WROWS * O) Xw[r * XS + (e - r * O)] = xr[u];
        }
        PROMP_STAMP(8 + 16 * (tix < 3 ? tix : 3) + 9);
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
        const bool rvalid = erow < nrows;
        const long long n = (long long)base + (rvalid ? erow : 0);
        const float* olsp = a.old_log_std + (a.ls_per_row ? n * A : (long long)task * A);
        // n is a valid row even for padding lanes; q0/q1 are valid action indices even for lanes that own none:
        // all loads are unconditional, the selects below discard what is not owned
        const float advn = rvalid ? a.adv[n] : 0.f;
        const float ac0 = a.act[n * A + q0], ac1 = a.act[n * A + q1];
        const float mo0 = a.old_mean[n * A + q0], mo1 = a.old_mean[n * A + q1];
        const float so0 = olsp[q0], so1 = olsp[q1];
        PROMP_STAMP(8 + 16 * (tix < 3 ? tix : 3) + 10);
        wave_sync();
        PROMP_STAMP(8 + 16 * (tix < 3 ? tix : 3) + 1);
        // ---- layer 1: H1 = tanh(X W1 + b1)
        {
            f32x4 acc[1][NC1];
#pragma unroll
            for (int j = 0; j < NC1; ++j) acc[0][j] = splat4(b1s[16 * j + i16]);   // bias rides in the accumulator
            outer16<1, NC1>(acc, Xw + i16 * XS + kk, 4, 0, W1s + kk * H1 + i16, 4 * H1, 16, Opad4 / 4, 1.f);
            PROMP_STAMP(8 + 16 * (tix < 3 ? tix : 3) + 11);
#pragma unroll
            for (int j = 0; j < NC1; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) H1w[(4 * kk + r) * HS + 16 * j + i16] = fast_tanh(acc[0][j][r]);
        }
        wave_sync();
        PROMP_STAMP(8 + 16 * (tix < 3 ? tix : 3) + 2);
        // ---- layer 2
        {
            f32x4 acc[1][NC2];
#pragma unroll
            for (int j = 0; j < NC2; ++j) acc[0][j] = splat4(b2s[16 * j + i16]);
            outer16<1, NC2>(acc, H1w + i16 * HS + kk * Q1, 1, 0, W2s + kk * Q1 * WS + i16, WS, 16, Q1, 1.f);
            PROMP_STAMP(8 + 16 * (tix < 3 ? tix : 3) + 12);
#pragma unroll
            for (int j = 0; j < NC2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) H2w[(4 * kk + r) * HS + 16 * j + i16] = fast_tanh(acc[0][j][r]);
        }
        wave_sync();
        PROMP_STAMP(8 + 16 * (tix < 3 ? tix : 3) + 3);
        // ---- output layer (16 padded columns)
        {
            f32x4 acc[1][1];
            acc[0][0] = splat4(b3s[i16]);
            outer16<1, 1>(acc, H2w + i16 * HS + kk * Q2, 1, 0, W3s + kk * Q2 * W3S + i16, W3S, 0, Q2, 1.f);
#pragma unroll
            for (int r = 0; r < 4; ++r) Msw[(4 * kk + r) * MS + i16] = acc[0][0][r];
        }
        wave_sync();
        PROMP_STAMP(8 + 16 * (tix < 3 ? tix : 3) + 4);
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
        PROMP_STAMP(8 + 16 * (tix < 3 ? tix : 3) + 5);
        if (BWD) {
        // ---- output-kernel gradient (+=); dZ2 = (dmu W3^T) * (1 - H2^2) in place over H2
        outer16<NC2, 1>(aw3, H2w + kk * HS + i16, 4 * HS, 16, Msw + kk * MS + i16, 4 * MS, 0, PROMP_WROWS / 4, 1.f);
        sched_fence();
        {
            f32x4 acc[1][NC2];
#pragma unroll
            for (int j = 0; j < NC2; ++j) acc[0][j] = zero4();
            outer16<1, NC2>(acc, Msw + i16 * MS + kk, 4, 0, W3Ts + kk * H2 + i16, 4 * H2, 16, 2, 1.f);
#pragma unroll
            for (int j = 0; j < NC2; ++j) {
                float cs = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = (4 * kk + r) * HS + 16 * j + i16;
                    const float h = H2w[idx];
                    const float d = acc[0][j][r] * (1.f - h * h);
                    H2w[idx] = d;
                    cs += d;
                }
                gb2[j] += cs;
            }
        }
        wave_sync();
        PROMP_STAMP(8 + 16 * (tix < 3 ? tix : 3) + 6);
        // ---- hidden_1 kernel gradient (+=); dZ1 = (dZ2 W2^T) * (1 - H1^2) in place over H1
        outer16<NC1, NC2>(aw2, H1w + kk * HS + i16, 4 * HS, 16, H2w + kk * HS + i16, 4 * HS, 16, PROMP_WROWS / 4, 1.f);
        sched_fence();
        {
            f32x4 acc[1][NC1];
#pragma unroll
            for (int j = 0; j < NC1; ++j) acc[0][j] = zero4();
            outer16<1, NC1>(acc, H2w + i16 * HS + kk * Q2, 1, 0, W2s + i16 * WS + kk * Q2, 1, 16 * WS, Q2, 1.f);
#pragma unroll
            for (int j = 0; j < NC1; ++j) {
                float cs = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = (4 * kk + r) * HS + 16 * j + i16;
                    const float h = H1w[idx];
                    const float d = acc[0][j][r] * (1.f - h * h);
                    H1w[idx] = d;
                    cs += d;
                }
                gb1[j] += cs;
            }
        }
        wave_sync();
        PROMP_STAMP(8 + 16 * (tix < 3 ? tix : 3) + 7);
        // ---- hidden_0 kernel gradient (+=): rows = observation index (two 16-blocks cover O <= 32)
        outer16<2, NC1>(aw1, Xw + kk * XS + i16, 4 * XS, 16, H1w + kk * HS + i16, 4 * HS, 16, PROMP_WROWS / 4, 1.f);
        wave_sync();
        }
        PROMP_STAMP(8 + 16 * (tix < 3 ? tix : 3) + 8);
    }
    PROMP_STAMP(2);

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
        __syncthreads();
        if (lane == 0) {
            SC[2 * w] = loss;
            SC[2 * w + 1] = klsum;
        }
        __syncthreads();
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
    float* S = sm;                                   // whole LDS allocation is free now
    const int NW2 = H1 * H2;                         // round 1: hidden_1 kernel
    const int NR2 = NP + 2 - NW2;                    // round 2: everything else, compacted
    PROMP_STAMP(120);
    __syncthreads();
    PROMP_STAMP(121);
    {
        float* mine = S + w * NW2;
#pragma unroll
        for (int i = 0; i < NC1; ++i)
#pragma unroll
            for (int j = 0; j < NC2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[(16 * i + 4 * kk + r) * H2 + 16 * j + i16] = aw2[i][j][r];
    }
    PROMP_STAMP(122);
    __syncthreads();
    PROMP_STAMP(123);
#pragma unroll 2
    for (int e = tid; e < NW2; e += NT) {
        float v[NW];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) v[ww] = S[ww * NW2 + e];      // all slab reads in flight together
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            if (ww < pw.nw0) t0 += v[ww];
            else t1 += v[ww];
        }
        P0[oW2 + e] = t0;
        if (two) P1[oW2 + e] = t1;
    }
    PROMP_STAMP(124);
    __syncthreads();
    PROMP_STAMP(125);
    {
        // compact index space of round 2: [0,oW2) hidden_0 kernel+bias | then everything after the hidden_1 kernel
        float* mine = S + w * NR2;
        for (int e = lane; e < NR2; e += 64) mine[e] = 0.f;
        wave_sync();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NC1; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * i + 4 * kk + r;
                    if (row < O) mine[row * H1 + 16 * j + i16] = aw1[i][j][r];
                }
#pragma unroll
        for (int j = 0; j < NC2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (i16 < A) mine[oW3 - NW2 + (16 * j + 4 * kk + r) * A + i16] = aw3[j][0][r];
        if (kk == 0) {
#pragma unroll
            for (int j = 0; j < NC1; ++j) mine[ob1 + 16 * j + i16] = gb1[j];
#pragma unroll
            for (int j = 0; j < NC2; ++j) mine[ob2 - NW2 + 16 * j + i16] = gb2[j];
        }
        if (lane < 4) {   // lane == q
            if (lane < A) {
                mine[ob3 - NW2 + lane] = gb30;
                mine[oS - NW2 + lane] = gs0 * lmask_reg0;
            }
            if (lane + 4 < A) {
                mine[ob3 - NW2 + lane + 4] = gb31;
                mine[oS - NW2 + lane + 4] = gs1 * lmask_reg1;
            }
        }
        if (lane == 0) {
            mine[NP - NW2] = loss;
            mine[NP + 1 - NW2] = klsum;
        }
    }
    PROMP_STAMP(126);
    __syncthreads();
    PROMP_STAMP(127);
    for (int e = tid; e < NR2; e += NT) {
        const int dst = e < oW2 ? e : e + NW2;
        float v[NW];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) v[ww] = S[ww * NR2 + e];
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            if (ww < pw.nw0) t0 += v[ww];
            else t1 += v[ww];
        }
        P0[dst] = t0;
        if (two) P1[dst] = t1;
    }
    PROMP_STAMP(4);
    if (PROMP_STAMPS_ON && a.dbg != nullptr && blockIdx.x == 0 && tid == 0)
        for (int i = 0; i < 128; ++i) a.dbg[i] = dbgs[i];
}

// ---------------------------------------------------------------------------------------------
// k_hvp:  out = -(d^2 L/d theta^2) v + kl_weight * grad KL   (R-operator, see oracle/promp.py:hvp)
// "q" quantities are (-R{.} + kl_weight * dKL{.}) of the reverse pass.
//
// Same wave-private structure as k_fwd_bwd (16-row tiles, 16x16x4 MFMA, no workgroup barrier in the tile loop);
// theta AND the direction v are staged in LDS (2 x 28.8 KB) so that no MFMA operand comes from global memory;
// each wave owns X, H1, RH1, H2, RH2 and two mean tiles (20.4 KB): 140 KB per workgroup, one workgroup per CU.
// dZ2 / qZ2 overwrite H2 / RH2 and qZ1 overwrites H1 in place.
// ---------------------------------------------------------------------------------------------
struct LdsHvp {
    int w1, b1, w2, b2, w3, w3t, b3, ls, lmask, es, sn2, vls;
    int vw1, vb1, vw2, vb2, vw3, vw3t, vb3;
    int wave0, wave_stride, x, h1, rh1, h2, rh2, ms, ms2;
    int total, HS, WS, Opad4, dbg;
};

PROMP_HD LdsHvp make_layout_hvp(int O, int H1, int H2, int nwaves, int NP) {
    LdsHvp L;
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
    PROMP_TAKE(vls, 16);
    PROMP_TAKE(vw1, L.Opad4 * H1);
    PROMP_TAKE(vb1, H1);
    PROMP_TAKE(vw2, H1 * L.WS);
    PROMP_TAKE(vb2, H2);
    PROMP_TAKE(vw3, H2 * PROMP_W3S);
    PROMP_TAKE(vw3t, 8 * H2);
    PROMP_TAKE(vb3, 16);
    L.HS = (H1 > H2 ? H1 : H2) + 1;
    L.wave0 = o;
    int q = 0;
    L.x = q;   q += (PROMP_WROWS * PROMP_XS + 3) & ~3;
    L.h1 = q;  q += (PROMP_WROWS * L.HS + 3) & ~3;
    L.rh1 = q; q += (PROMP_WROWS * L.HS + 3) & ~3;
    L.h2 = q;  q += (PROMP_WROWS * L.HS + 3) & ~3;
    L.rh2 = q; q += (PROMP_WROWS * L.HS + 3) & ~3;
    L.ms = q;  q += (PROMP_WROWS * PROMP_MS + 3) & ~3;
    L.ms2 = q; q += (PROMP_WROWS * PROMP_MS + 3) & ~3;
    L.wave_stride = q;
    o += nwaves * q;
    {
        const int nw2 = H1 * H2, nr2 = NP + 2 - nw2;
        const int need = 4 * (nw2 > nr2 ? nw2 : nr2);   // the end-of-kernel slabs start at 0
        if (o < need) o = need;
    }
    L.dbg = o;
    o += 256;
#undef PROMP_TAKE
    L.total = o;
    return L;
}

// one task's network weights -> LDS (kernel layouts of make_layout_*); src is a flat [Theta] vector
template <int H1, int H2>
PROMP_DEV void stage_net(float* W1s, float* b1s, float* W2s, float* b2s, float* W3s, float* W3Ts, float* b3s,
                         const float* src, int O, int A, int Opad4, int WS, int tid) {
    const int ob1 = O * H1, oW2 = ob1 + H1, ob2 = oW2 + H1 * H2, oW3 = ob2 + H2, ob3 = oW3 + H2 * A;
    // every global load is issued before the first LDS store: one L2 round trip, not one per loop iteration
    constexpr int NT = 256, N1 = (32 * H1 + NT - 1) / NT, N2 = H1 * H2 / NT, N3 = (H2 * 16 + NT - 1) / NT, N3T = (8 * H2 + NT - 1) / NT;
    float r1[N1], r2[N2], r3[N3], r3t[N3T];
#pragma unroll
    for (int i = 0; i < N1; ++i) {
        const int e = tid + i * NT;
        r1[i] = (e < O * H1) ? src[e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < N2; ++i) r2[i] = src[oW2 + tid + i * NT];
#pragma unroll
    for (int i = 0; i < N3; ++i) {
        const int e = tid + i * NT, k = e >> 4, j = e & 15;
        r3[i] = (e < H2 * 16 && j < A) ? src[oW3 + k * A + j] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < N3T; ++i) {
        const int e = tid + i * NT, aa = e / H2, k = e - aa * H2;
        r3t[i] = (e < 8 * H2 && aa < A) ? src[oW3 + k * A + aa] : 0.f;
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
        if (e < H2 * 16) W3s[k * PROMP_W3S + j] = r3[i];
    }
#pragma unroll
    for (int i = 0; i < N3T; ++i) {
        const int e = tid + i * NT;
        if (e < 8 * H2) W3Ts[e] = r3t[i];
    }
    if (tid < H1) b1s[tid] = src[ob1 + tid];
    if (tid < H2) b2s[tid] = src[ob2 + tid];
    if (tid < 16) b3s[tid] = (tid < A) ? src[ob3 + tid] : 0.f;
}

template <int NB1, int NB2>
__global__ void __launch_bounds__(256, 1) k_hvp(PassArgs a) {
    constexpr int H1 = 32 * NB1, H2 = 32 * NB2, NC1 = H1 / 16, NC2 = H2 / 16, MS = PROMP_MS, XS = PROMP_XS, W3S = PROMP_W3S;
    constexpr int Q1 = H1 / 4, Q2 = H2 / 4;
    PROMP_SMEM_DECL;
    float* sm = (float*)PROMP_SMEM_PTR;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i16 = lane & 15, kk = lane >> 4;
    const WorkItem wk = a.work[blockIdx.x];
    const int task = wk.task;
    const int O = a.O, A = a.A;
    const int ob1 = O * H1, oW2 = ob1 + H1, ob2 = oW2 + H1 * H2, oW3 = ob2 + H2, ob3 = oW3 + H2 * A, oS = ob3 + A,
              NP = oS + A;
    // the layout is sized for obs_dim 32 whatever O is: every LDS offset is then a compile-time constant and folds into
    // the ds_read / ds_write immediates instead of costing address arithmetic in the tile loop
    const LdsHvp L = make_layout_hvp(32, H1, H2, 4, 0);
    const int HS = L.HS, WS = L.WS, Opad4 = (O + 3) & ~3;
    float *W1s = sm + L.w1, *b1s = sm + L.b1, *W2s = sm + L.w2, *b2s = sm + L.b2, *W3s = sm + L.w3,
          *W3Ts = sm + L.w3t, *b3s = sm + L.b3, *lss = sm + L.ls, *lmask = sm + L.lmask, *ess = sm + L.es,
          *sn2s = sm + L.sn2, *vls = sm + L.vls;
    float *vW1s = sm + L.vw1, *vb1s = sm + L.vb1, *vW2s = sm + L.vw2, *vb2s = sm + L.vb2, *vW3s = sm + L.vw3,
          *vW3Ts = sm + L.vw3t, *vb3s = sm + L.vb3;
    unsigned long long* dbgs = (unsigned long long*)(sm + L.dbg);
    if (PROMP_STAMPS_ON && a.dbg != nullptr && blockIdx.x == 0 && tid < 128) dbgs[tid] = 0;
    float* wreg = sm + L.wave0 + w * L.wave_stride;
    float *Xw = wreg + L.x, *H1w = wreg + L.h1, *RH1w = wreg + L.rh1, *H2w = wreg + L.h2, *RH2w = wreg + L.rh2,
          *Msw = wreg + L.ms, *Ms2w = wreg + L.ms2;
    const int ntask = a.task_row_offsets[task + 1] - a.task_row_offsets[task];
    const float invN = 1.0f / (float)ntask;
    const float* th = a.theta + (long long)task * a.theta_task_stride;
    const float* v = a.vdir + (long long)task * NP;
    PROMP_STAMP(0);

    stage_net<H1, H2>(W1s, b1s, W2s, b2s, W3s, W3Ts, b3s, th, O, A, Opad4, WS, tid);
    stage_net<H1, H2>(vW1s, vb1s, vW2s, vb2s, vW3s, vW3Ts, vb3s, v, O, A, Opad4, WS, tid);
    if (tid < 16) {
        const float sr = (tid < A) ? th[oS + tid] : 0.f;
        const bool clipped = a.clip_log_std && (sr < a.min_log_std);
        const float s = clipped ? a.min_log_std : sr;
        lss[tid] = s;
        lmask[tid] = clipped ? 0.f : 1.f;
        ess[tid] = expf(-s);
        sn2s[tid] = expf(2.f * s);
        vls[tid] = (tid < A && !clipped) ? v[oS + tid] : 0.f;   // R{s} = mask * v_s
    }
    for (int e = lane; e < PROMP_WROWS * XS; e += 64) Xw[e] = 0.f;
    __syncthreads();
    PROMP_STAMP(1);

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
    float ob1acc[NC1], ob2acc[NC2];
#pragma unroll
    for (int j = 0; j < NC1; ++j) ob1acc[j] = 0.f;
#pragma unroll
    for (int j = 0; j < NC2; ++j) ob2acc[j] = 0.f;
    float klsum = 0.f, outs0 = 0.f, outs1 = 0.f, outb30 = 0.f, outb31 = 0.f;
    const float klw = a.kl_weight;

    const int erow = lane >> 2, q = lane & 3;
    const bool own0 = q < A, own1 = (q + 4) < A;
    const int q0 = own0 ? q : 0, q1 = own1 ? q + 4 : 0;   // clamped action indices for unconditional loads
    int xoff[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = lane + 64 * u;
        xoff[u] = (e < PROMP_WROWS * O) ? (e / O) * XS + (e % O) : -1;
    }
    const int first = wk.row_begin + PROMP_WROWS * w;
    float xr[8];
    {
        const int nr = (wk.row_end - first) < PROMP_WROWS ? (wk.row_end - first) : PROMP_WROWS;
        const int lim = (first < wk.row_end) ? nr * O : 0;
        const float* src = a.obs + (long long)(first < wk.row_end ? first : wk.row_begin) * O;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = lane + 64 * u;
            const float x = src[e < lim ? e : 0];      // always a valid address: no exec-masked branch per load
            xr[u] = (e < lim) ? x : 0.f;
        }
    }

    for (int base = first; base < wk.row_end; base += 4 * PROMP_WROWS) {
        const int nrows = (wk.row_end - base) < PROMP_WROWS ? (wk.row_end - base) : PROMP_WROWS;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (xoff[u] >= 0) Xw[xoff[u]] = xr[u];
        {
            const int nb = base + 4 * PROMP_WROWS;
            const int nn = (wk.row_end - nb) < PROMP_WROWS ? (wk.row_end - nb) : PROMP_WROWS;
            const int lim = (nb < wk.row_end) ? nn * O : 0;
            const float* src = a.obs + (long long)(nb < wk.row_end ? nb : wk.row_begin) * O;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = lane + 64 * u;
                const float x = src[e < lim ? e : 0];
                xr[u] = (e < lim) ? x : 0.f;
            }
        }
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
        // ---- layer 1 and its tangent:  Rz1 = X vW1 + vb1
        {
            f32x4 az[1][NC1], ar[1][NC1];
#pragma unroll
            for (int j = 0; j < NC1; ++j) {
                az[0][j] = splat4(b1s[16 * j + i16]);     // biases ride in the accumulators
                ar[0][j] = splat4(vb1s[16 * j + i16]);
            }
            outer16_pt<1, NC1, false>(az, ar, Xw + i16 * XS + kk, nullptr, 4, 0, W1s + kk * H1 + i16, vW1s + kk * H1 + i16, 4 * H1, 16,
                                      Opad4 / 4, 1.f);
#pragma unroll
            for (int j = 0; j < NC1; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = (4 * kk + r) * HS + 16 * j + i16;
                    const float h = fast_tanh(az[0][j][r]);
                    H1w[idx] = h;
                    RH1w[idx] = (1.f - h * h) * ar[0][j][r];
                }
        }
        wave_sync();
        // ---- layer 2 and its tangent:  Rz2 = H1 vW2 + RH1 W2 + vb2
        {
            f32x4 az[1][NC2], ar[1][NC2];
#pragma unroll
            for (int j = 0; j < NC2; ++j) {
                az[0][j] = splat4(b2s[16 * j + i16]);
                ar[0][j] = splat4(vb2s[16 * j + i16]);
            }
            outer16_pt<1, NC2, true>(az, ar, H1w + i16 * HS + kk * Q1, RH1w + i16 * HS + kk * Q1, 1, 0, W2s + kk * Q1 * WS + i16,
                                     vW2s + kk * Q1 * WS + i16, WS, 16, Q1, 1.f);
#pragma unroll
            for (int j = 0; j < NC2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = (4 * kk + r) * HS + 16 * j + i16;
                    const float h = fast_tanh(az[0][j][r]);
                    H2w[idx] = h;
                    RH2w[idx] = (1.f - h * h) * ar[0][j][r];
                }
        }
        wave_sync();
        // ---- output layer and its tangent:  Rmu = H2 vW3 + RH2 W3 + vb3
        {
            f32x4 am[1][1], ar[1][1];
            am[0][0] = splat4(b3s[i16]);
            ar[0][0] = splat4(vb3s[i16]);
            outer16_pt<1, 1, true>(am, ar, H2w + i16 * HS + kk * Q2, RH2w + i16 * HS + kk * Q2, 1, 0, W3s + kk * Q2 * W3S + i16,
                                   vW3s + kk * Q2 * W3S + i16, W3S, 0, Q2, 1.f);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Msw[(4 * kk + r) * MS + i16] = am[0][0][r];
                Ms2w[(4 * kk + r) * MS + i16] = ar[0][0][r];
            }
        }
        wave_sync();
        // ---- loss-level R-operator: 4 lanes per row, each owns actions {q, q+4}
        {
            float dlp = 0.f, Rlp = 0.f, kl = 0.f;
            float z0 = 0.f, z1 = 0.f, e0 = 0.f, e1 = 0.f, Rmu0 = 0.f, Rmu1 = 0.f, dklm0 = 0.f, dklm1 = 0.f, dkls0 = 0.f,
                  dkls1 = 0.f, Rs0 = 0.f, Rs1 = 0.f;
            if (own0) {
                const float s = lss[q], mu = Msw[erow * MS + q];
                Rmu0 = Ms2w[erow * MS + q];
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
            }
            if (own1) {
                const float s = lss[q + 4], mu = Msw[erow * MS + q + 4];
                Rmu1 = Ms2w[erow * MS + q + 4];
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
            if (own0) {
                const float Rz = -Rmu0 * e0 - z0 * Rs0;
                const float d = c * z0 * e0;
                const float Rd = Rc * z0 * e0 + c * (Rz * e0 - z0 * e0 * Rs0);
                const float Rds = Rc * (z0 * z0 - 1.f) + 2.f * c * z0 * Rz;
                const float qm = km * (-Rd + klw * dklm0);
                Msw[erow * MS + q] = d;
                Ms2w[erow * MS + q] = qm;
                outs0 += km * (-Rds + klw * dkls0);
                outb30 += qm;
            }
            if (own1) {
                const float Rz = -Rmu1 * e1 - z1 * Rs1;
                const float d = c * z1 * e1;
                const float Rd = Rc * z1 * e1 + c * (Rz * e1 - z1 * e1 * Rs1);
                const float Rds = Rc * (z1 * z1 - 1.f) + 2.f * c * z1 * Rz;
                const float qm = km * (-Rd + klw * dklm1);
                Msw[erow * MS + q + 4] = d;
                Ms2w[erow * MS + q + 4] = qm;
                outs1 += km * (-Rds + klw * dkls1);
                outb31 += qm;
            }
            // columns >= A of Msw / Ms2w already hold exact zeros (zero-padded W3 / vW3 / biases)
        }
        wave_sync();
        // ---- out_W3 += -RH2^T dmu + H2^T qmu ; dZ2 over H2, qZ2 over RH2
        outer16_two<NC2, 1>(aw3, RH2w + kk * HS + i16, H2w + kk * HS + i16, 4 * HS, 16, Msw + kk * MS + i16, Ms2w + kk * MS + i16,
                            4 * MS, 0, PROMP_WROWS / 4, -1.f);
        {
            f32x4 ad[1][NC2], aq[1][NC2];
#pragma unroll
            for (int j = 0; j < NC2; ++j) ad[0][j] = aq[0][j] = zero4();
            outer16_pt<1, NC2, true>(ad, aq, Msw + i16 * MS + kk, Ms2w + i16 * MS + kk, 4, 0, W3Ts + kk * H2 + i16, vW3Ts + kk * H2 + i16,
                                     4 * H2, 16, 2, -1.f);
#pragma unroll
            for (int j = 0; j < NC2; ++j) {
                float cs = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = (4 * kk + r) * HS + 16 * j + i16;
                    const float h = H2w[idx], rh = RH2w[idx];
                    const float d1 = 1.f - h * h;
                    const float qz = aq[0][j][r] * d1 + 2.f * ad[0][j][r] * h * rh;
                    H2w[idx] = ad[0][j][r] * d1;
                    RH2w[idx] = qz;
                    cs += qz;
                }
                ob2acc[j] += cs;
            }
        }
        wave_sync();
        // ---- out_W2 += -RH1^T dZ2 + H1^T qZ2 ; qZ1 over H1
        outer16_two<NC1, NC2>(aw2, RH1w + kk * HS + i16, H1w + kk * HS + i16, 4 * HS, 16, H2w + kk * HS + i16, RH2w + kk * HS + i16,
                              4 * HS, 16, PROMP_WROWS / 4, -1.f);
        {
            f32x4 ad[1][NC1], aq[1][NC1];
#pragma unroll
            for (int j = 0; j < NC1; ++j) ad[0][j] = aq[0][j] = zero4();
            outer16_pt<1, NC1, true>(ad, aq, H2w + i16 * HS + kk * Q2, RH2w + i16 * HS + kk * Q2, 1, 0, W2s + i16 * WS + kk * Q2,
                                     vW2s + i16 * WS + kk * Q2, 1, 16 * WS, Q2, -1.f);
#pragma unroll
            for (int j = 0; j < NC1; ++j) {
                float cs = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = (4 * kk + r) * HS + 16 * j + i16;
                    const float h = H1w[idx], rh = RH1w[idx];
                    const float qz = aq[0][j][r] * (1.f - h * h) + 2.f * ad[0][j][r] * h * rh;
                    H1w[idx] = qz;
                    cs += qz;
                }
                ob1acc[j] += cs;
            }
        }
        wave_sync();
        // ---- out_W1 += X^T qZ1
        outer16<2, NC1>(aw1, Xw + kk * XS + i16, 4 * XS, 16, H1w + kk * HS + i16, 4 * HS, 16, PROMP_WROWS / 4, 1.f);
        wave_sync();
    }
    PROMP_STAMP(2);

#pragma unroll
    for (int j = 0; j < NC1; ++j) {
        ob1acc[j] += shfl_xor_f32(ob1acc[j], 16);
        ob1acc[j] += shfl_xor_f32(ob1acc[j], 32);
    }
#pragma unroll
    for (int j = 0; j < NC2; ++j) {
        ob2acc[j] += shfl_xor_f32(ob2acc[j], 16);
        ob2acc[j] += shfl_xor_f32(ob2acc[j], 32);
    }
#pragma unroll
    for (int m = 4; m <= 32; m <<= 1) {
        outs0 += shfl_xor_f32(outs0, m);  outs1 += shfl_xor_f32(outs1, m);  outb30 += shfl_xor_f32(outb30, m);
        outb31 += shfl_xor_f32(outb31, m);  klsum += shfl_xor_f32(klsum, m);
    }
    const float lmask_reg0 = lmask[lane & 3], lmask_reg1 = lmask[(lane & 3) + 4];
    float* P = a.partials + (long long)blockIdx.x * a.partial_stride;
    float* S = sm;
    const int NW2 = H1 * H2;
    const int NR2 = NP + 2 - NW2;
    __syncthreads();
    {
        float* mine = S + w * NW2;
#pragma unroll
        for (int i = 0; i < NC1; ++i)
#pragma unroll
            for (int j = 0; j < NC2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[(16 * i + 4 * kk + r) * H2 + 16 * j + i16] = aw2[i][j][r];
    }
    __syncthreads();
    for (int e = tid; e < NW2; e += 256) P[oW2 + e] = ((S[e] + S[NW2 + e]) + S[2 * NW2 + e]) + S[3 * NW2 + e];
    __syncthreads();
    {
        float* mine = S + w * NR2;
        for (int e = lane; e < NR2; e += 64) mine[e] = 0.f;
        wave_sync();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NC1; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * i + 4 * kk + r;
                    if (row < O) mine[row * H1 + 16 * j + i16] = aw1[i][j][r];
                }
#pragma unroll
        for (int j = 0; j < NC2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (i16 < A) mine[oW3 - NW2 + (16 * j + 4 * kk + r) * A + i16] = aw3[j][0][r];
        if (kk == 0) {
#pragma unroll
            for (int j = 0; j < NC1; ++j) mine[ob1 + 16 * j + i16] = ob1acc[j];
#pragma unroll
            for (int j = 0; j < NC2; ++j) mine[ob2 - NW2 + 16 * j + i16] = ob2acc[j];
        }
        if (lane < 4) {
            if (lane < A) {
                mine[ob3 - NW2 + lane] = outb30;
                mine[oS - NW2 + lane] = outs0 * lmask_reg0;
            }
            if (lane + 4 < A) {
                mine[ob3 - NW2 + lane + 4] = outb31;
                mine[oS - NW2 + lane + 4] = outs1 * lmask_reg1;
            }
        }
        if (lane == 0) {
            mine[NP - NW2] = 0.f;
            mine[NP + 1 - NW2] = klsum;
        }
    }
    __syncthreads();
    for (int e = tid; e < NR2; e += 256) {
        const float vv = ((S[e] + S[NR2 + e]) + S[2 * NR2 + e]) + S[3 * NR2 + e];
        P[e < oW2 ? e : e + NW2] = vv;
    }
    PROMP_STAMP(4);
    if (PROMP_STAMPS_ON && a.dbg != nullptr && blockIdx.x == 0 && tid == 0)
        for (int i = 0; i < 128; ++i) a.dbg[i] = dbgs[i];
}

// ---------------------------------------------------------------------------------------------
// Reductions over the per-workgroup partials of each task (fixed order => reproducible).
// grid = (ceil((NP+EXTRA)/256), n_tasks)
// ---------------------------------------------------------------------------------------------
struct ReduceArgs {
    const float* partials;
    int partial_stride;
    const int* task_wg_offsets;  // [tasks+1]: workgroups of task i are [o[i], o[i+1])
    int NP;                      // Theta
    const float* step_sizes;     // [Theta]
    // mode 0 (inner step): next[i] = cur[i] - alpha * g ; scal[i] = {loss, kl}
    // mode 1 (outer)     : lam[i] = g ; v[i] = alpha * g ; scal[i] = {loss, kl}
    // mode 2 (hvp)       : lam[i] += g ; v[i] = alpha * lam[i] ; scal[i] = {-, kl}
    // mode 3 (plain)     : lam[i] = g ; scal
    // mode 4 (scalars)   : scal only (forward-only passes write no gradient)
    int mode;
    const float* cur;            // [Theta] or [tasks][Theta]
    long long cur_task_stride;
    float* next;                 // [tasks][Theta]
    float* lam;                  // [tasks][Theta]
    float* v;                    // [tasks][Theta]
    float* scal;                 // [tasks][2]
};

__global__ void __launch_bounds__(256) k_reduce_task(ReduceArgs a) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int task = blockIdx.y;
    if (j >= a.NP + 2) return;
    if (a.mode == 4 && j < a.NP) return;
    float g = 0.f;
    const int wg0 = a.task_wg_offsets[task], wg1 = a.task_wg_offsets[task + 1];
#pragma unroll 4
    for (int wg = wg0; wg < wg1; ++wg) g += a.partials[(long long)wg * a.partial_stride + j];
    if (j >= a.NP) {
        a.scal[task * 2 + (j - a.NP)] = g;
        return;
    }
    const long long tj = (long long)task * a.NP + j;
    if (a.mode == 0) {
        a.next[tj] = a.cur[(long long)task * a.cur_task_stride + j] - a.step_sizes[j] * g;
        return;
    }
    float lam = g;
    if (a.mode == 2) lam += a.lam[tj];
    a.lam[tj] = lam;
    if (a.mode == 3) return;
    a.v[tj] = a.step_sizes[j] * lam;
}

// Task sum of lam (gradient) and of the per-task scalars -> red[NP + K + 2]
//   red[0..NP)      = sum_i lam[i][j]
//   red[NP]         = sum_i J_i                     (outer surrogate)
//   red[NP+1+k]     = sum_i KL^k_i
//   red[NP+1+K]     = sum_i outer KL_i
// grid = ceil((NP+K+2)/256)
struct FinalArgs {
    const float* lam;
    int NP, K, n_tasks;
    const float* scal_inner;  // [K][tasks][2]
    const float* scal_outer;  // [tasks][2]
    float* red;
    int want_grad;
};

// grid = ceil((NP + K + 2) / 64), block = 256: 64 columns x 4 task quarters.  Thread (c, q) adds the tasks i = q, q+4, ...
// of column c (10 dependent-latency steps instead of 40 at M = 40); the four quarter sums are added in a fixed order.
__global__ void __launch_bounds__(256) k_reduce_final(FinalArgs a) {
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c;
    float s = 0.f;
    if (j < a.NP) {
        if (a.want_grad) {
#pragma unroll 4
            for (int i = q; i < a.n_tasks; i += 4) s += a.lam[(long long)i * a.NP + j];
        }
    } else if (j == a.NP) {
        for (int i = q; i < a.n_tasks; i += 4) s += a.scal_outer[i * 2 + 0];
    } else if (j <= a.NP + a.K) {
        const int k = j - a.NP - 1;
        for (int i = q; i < a.n_tasks; i += 4) s += a.scal_inner[((long long)k * a.n_tasks + i) * 2 + 1];
    } else if (j == a.NP + a.K + 1) {
        for (int i = q; i < a.n_tasks; i += 4) s += a.scal_outer[i * 2 + 1];
    }
    part[q][c] = s;
    __syncthreads();
    if (q == 0 && j < a.NP + a.K + 2) a.red[j] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

// Mean over the global meta-batch + Adam.  red holds SUMS over all tasks (after the all-reduce).
// stats[0] = loss = mean_i J_i + mean_k(eta_k * mean_i KL^k_i); stats[1+k] = inner_kl[k]; stats[1+K] = outer_kl
struct AdamArgs {
    float* theta;
    float* m;
    float* v;
    const float* red;
    float* grad_mean;  // [NP] task-mean gradient (kept for promp_meta_grad's output)
    float* stats;      // [K+2]
    const float* eta;  // [K]
    int NP, K;
    float inv_tasks;
    float lr_t;        // lr * sqrt(1-b2^t)/(1-b1^t); 0 => no parameter update (stats / grad only)
    int do_update;
};

__global__ void __launch_bounds__(256) k_mean_adam(AdamArgs a) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < a.NP) {
        const float g = a.red[j] * a.inv_tasks;
        a.grad_mean[j] = g;
        if (a.do_update) {
            const float m = 0.9f * a.m[j] + 0.1f * g;
            const float v = 0.999f * a.v[j] + 0.001f * g * g;
            a.m[j] = m;
            a.v[j] = v;
            a.theta[j] -= a.lr_t * m / (sqrtf(v) + 1e-8f);
        }
    } else if (j == a.NP) {
        float pen = 0.f;
        for (int k = 0; k < a.K; ++k) {
            const float ikl = a.red[a.NP + 1 + k] * a.inv_tasks;
            a.stats[1 + k] = ikl;
            pen += a.eta[k] * ikl;
        }
        a.stats[0] = a.red[a.NP] * a.inv_tasks + pen / (float)a.K;
        a.stats[1 + a.K] = a.red[a.NP + 1 + a.K] * a.inv_tasks;
    }
}

// k_reduce_final + k_mean_adam in one launch, for the single-rank case (no all-reduce in between).  Column sums in the
// same order as k_reduce_final (bitwise the same red[]); the thread that finishes a parameter column applies its Adam
// update; one extra workgroup re-sums the K + 2 scalar columns and writes the statistics.
// grid = ceil((NP + K + 2) / 64) + 1, block = 256
__global__ void __launch_bounds__(256) k_final_adam(FinalArgs a, AdamArgs ad) {
    __shared__ float part[4][64];
    __shared__ float sums[64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const bool stats_block = blockIdx.x == gridDim.x - 1;
    const int j = stats_block ? a.NP + c : blockIdx.x * 64 + c;
    float s = 0.f;
    if (j < a.NP) {
        if (a.want_grad) {
#pragma unroll 4
            for (int i = q; i < a.n_tasks; i += 4) s += a.lam[(long long)i * a.NP + j];
        }
    } else if (j == a.NP) {
        for (int i = q; i < a.n_tasks; i += 4) s += a.scal_outer[i * 2 + 0];
    } else if (j <= a.NP + a.K) {
        const int k = j - a.NP - 1;
        for (int i = q; i < a.n_tasks; i += 4) s += a.scal_inner[((long long)k * a.n_tasks + i) * 2 + 1];
    } else if (j == a.NP + a.K + 1) {
        for (int i = q; i < a.n_tasks; i += 4) s += a.scal_outer[i * 2 + 1];
    }
    part[q][c] = s;
    __syncthreads();
    if (q == 0 && j < a.NP + a.K + 2) {
        const float t = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
        if (!stats_block) a.red[j] = t;
        sums[c] = t;
        if (!stats_block && j < a.NP) {
            const float g = t * ad.inv_tasks;
            ad.grad_mean[j] = g;
            if (ad.do_update) {
                const float m = 0.9f * ad.m[j] + 0.1f * g;
                const float v = 0.999f * ad.v[j] + 0.001f * g * g;
                ad.m[j] = m;
                ad.v[j] = v;
                ad.theta[j] -= ad.lr_t * m / (sqrtf(v) + 1e-8f);
            }
        }
    }
    if (!stats_block) return;
    __syncthreads();
    if (threadIdx.x == 0) {      // sums[0] = J, sums[1..K] = inner KLs, sums[K+1] = outer KL (sums over the tasks)
        float pen = 0.f;
        for (int k = 0; k < a.K; ++k) {
            const float ikl = sums[1 + k] * ad.inv_tasks;
            ad.stats[1 + k] = ikl;
            pen += ad.eta[k] * ikl;
        }
        ad.stats[0] = sums[0] * ad.inv_tasks + pen / (float)a.K;
        ad.stats[1 + a.K] = sums[1 + a.K] * ad.inv_tasks;
    }
}

// dst[i][:] = src[:] for i < n_tasks   (MetaPolicy.switch_to_pre_update)
__global__ void __launch_bounds__(256) k_replicate(float* dst, const float* src, int NP, int n_tasks) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < NP)
        for (int i = 0; i < n_tasks; ++i) dst[(long long)i * NP + j] = src[j];
}

// ---------------------------------------------------------------------------------------------
// k_policy_forward: mean network of every task's CURRENT parameters on a small batch of observations
// (MetaGaussianMLPPolicy.get_actions, policies/meta_gaussian_mlp_policy.py:99-157: [tasks][B][O] -> [tasks][B][A]).
// Rollout-time inference is a few hundred rows per environment step: one workgroup per task, one thread per row,
// weights read through the scalar/L1 path; it is latency-, not throughput-bound.
// grid = tasks, block = 256
// ---------------------------------------------------------------------------------------------
struct ForwardArgs {
    const float* obs;          // [tasks][B][O]
    const float* theta_tasks;  // [tasks][Theta]
    float* mean;               // [tasks][B][A]
    int B, O, A, H1, H2;
};

__global__ void __launch_bounds__(256) k_policy_forward(ForwardArgs a) {
    const int task = blockIdx.x;
    const int O = a.O, A = a.A, H1 = a.H1, H2 = a.H2;
    const int ob1 = O * H1, oW2 = ob1 + H1, ob2 = oW2 + H1 * H2, oW3 = ob2 + H2, ob3 = oW3 + H2 * A, NP = ob3 + 2 * A;
    const float* th = a.theta_tasks + (long long)task * NP;
    for (int row = threadIdx.x; row < a.B; row += 256) {
        const float* x = a.obs + ((long long)task * a.B + row) * O;
        float h1[128], h2[128];   // hidden sizes up to 128
        for (int j = 0; j < H1; ++j) {
            float z = th[ob1 + j];
            for (int k = 0; k < O; ++k) z = fmaf(x[k], th[k * H1 + j], z);
            h1[j] = fast_tanh(z);
        }
        for (int j = 0; j < H2; ++j) {
            float z = th[ob2 + j];
            for (int k = 0; k < H1; ++k) z = fmaf(h1[k], th[oW2 + k * H2 + j], z);
            h2[j] = fast_tanh(z);
        }
        for (int j = 0; j < A; ++j) {
            float z = th[ob3 + j];
            for (int k = 0; k < H2; ++k) z = fmaf(h2[k], th[oW3 + k * A + j], z);
            a.mean[((long long)task * a.B + row) * A + j] = z;
        }
    }
}
