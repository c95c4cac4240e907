// promp_kernels_pass2.h -- k_pass_pair: the first-order policy pass (objective, mean KL, gradient: reference rows a8-a11) for the
// (64, 64) network, TWO waves per SIMD.
//
// Same arithmetic, arguments, segment table, partial rows and primal-cache blocks as k_pass (promp_kernels_pass.h: every GEMM on
// the BF16 matrix pipe in float32-equivalent arithmetic, six products of a three-way split, register-chained transposed layers,
// weight gradients through transposed LDS tiles and ds_read_b64_tr_b16).  What changes is who does the work.
//
// k_pass runs one wave per SIMD (512 registers: 144 of them gradient accumulators), and a lone wave issues in order: a vector
// instruction every ~4 cycles at best, every dependent chain (the three-way splits: 11 dependent instructions per value pair)
// exposed.  Measured on this chip (profiles/r02_bf16_issue_microbench.txt): the split of a value pair costs 37 ns on one wave per
// SIMD and 23 ns for TWO pairs on two waves -- 3.2x.  k_pass spends 7 k of its 9.6 k cycles per tile in such vector regions.
//
// Here a 16-sample tile belongs to a PAIR of waves, and each wave owns HALF the units of every layer (half h owns the 16-unit
// blocks 2h, 2h + 1 = the k-chunk P = h of the next layer's BF16 instruction).  Everything that scales with units halves per wave:
// matrix instructions, tanh, splits, weight fragments in flight, and the gradient accumulators (the hidden_1 kernel gradient's
// columns, the hidden_0 kernel gradient's columns, the output kernel gradient's rows of the wave's own units) -- so a wave fits
// in 256 registers and every SIMD runs two.  What the halves owe each other goes through the pair's LDS tiles, which the weight
// gradients need written anyway:
//   * layer 2 needs all of H1: the partner's BF16 planes are read back from the transposed tile in chain order (the addresses
//     the partner wrote them at: 2 x 8 bytes per plane) -- after a workgroup barrier (B1);
//   * the means: each half contracts its own 32 hidden_1 units, the two partial means cross through 512 bytes of LDS (B2), both
//     halves then run the (cheap, per-row) distribution epilogue -- no third exchange for the mean cotangents;
//   * dH1 = W2 dZ2^T needs all of dZ2: as for layer 2 (B3).
// Each half starts a contraction with its OWN chunk before the barrier, so the matrix pipe works while the partner arrives.
// The weight gradients need no exchange beyond the tiles: aw2[:, own] = H1^T dZ2[own] reads both halves' H1 planes (transpose
// reads issued before B3, while the partner's hidden_0 planes are still there), everything else is the wave's own data.
// Three workgroup barriers per tile; every LDS hazard between the halves is separated by one of them (noted at each use).
// The barriers are the workgroup's (gfx950 has no named barriers): the four pairs walk their tiles in lockstep, a pair past the
// end of a segment walks a null tile (zero weights, no stores).
#pragma once
#include "promp_kernels_pass.h"

struct Pass2Lds {                // offsets in 4-byte words
    int pair0, pair_stride;      // per pair: the two transposed tiles, the mean exchange, the cotangent tile, an observation tile per half
    int ta, tb, mu, dm, xt0, xt_stride, flags;
    int ctab, dtab;              // per-lane constants read once per tile instead of living in registers: addresses of the
                                 // cotangent-tile accesses [lane][8]; distribution parameters of the lane's two actions [lane][8]
    int total;
};
#define PROMP_PASS2_PAIRS 4
PROMP_CX Pass2Lds pass2_layout(int NP) {
    Pass2Lds Q{};
    const PassLds L = pass_layout(4, 4, 0, 0);
    Q.pair0 = L.wave0;
    int q = 0;
    Q.ta = q; q += 3 * PROMP_PASS_TPLANE;
    Q.tb = q; q += 3 * PROMP_PASS_TPLANE;
    Q.mu = q; q += 256;                                  // [half 2][lane 64][2]
    Q.dm = q; q += 3 * PROMP_PASS_DPLANE;                // both halves write the same values (each computes every mean cotangent)
    Q.xt0 = q; Q.xt_stride = 3 * PROMP_PASS_XPLANE; q += 2 * Q.xt_stride;   // private: read back by the writer only, no barrier involved
    Q.flags = q; q += 4;                                 // rendezvous words of the two halves
    Q.pair_stride = q;
    int o = Q.pair0 + PROMP_PASS2_PAIRS * q;
    Q.ctab = o; o += 512;
    Q.dtab = o; o += 512;
    {   // end of segment: one slab of [NP + 2] floats per pair, from offset 4 (aliases everything else)
        const int need = 4 + PROMP_PASS2_PAIRS * ((NP + 2 + 3) & ~3);
        if (o < need) o = need;
    }
    Q.total = o;
    return Q;
}

struct Pass2Sums {               // what a wave (half h of its pair) accumulates over its tiles of a segment
    f32x16 aw2[2];               // hidden_1 kernel gradient: rows 32 bi .. + 31 (hidden_0 units), columns 32 h .. + 31
    f32x16 aw1;                  // hidden_0 kernel gradient: rows = observation slots, columns 32 h .. + 31
    f32x4 aw3[2], gb1[2], gb2[2];   // blocks 2 h + j
    float loss, klsum, gs0, gs1, gb30, gb31;       // half 0 only
};

// the value of the launch's objective kind (exactly one mask is all ones): no select, no branch
PROMP_DEV float pass2_pick(const PassWalk& W, float v_kl, float v_ratio, float v_clip, float v_ll) {
    const unsigned b = (__builtin_bit_cast(unsigned, v_kl) & W.m_kl) | (__builtin_bit_cast(unsigned, v_ratio) & W.m_ratio) |
                       (__builtin_bit_cast(unsigned, v_clip) & W.m_clip) | (__builtin_bit_cast(unsigned, v_ll) & W.m_ll);
    return __builtin_bit_cast(float, b);
}
// rendezvous of the two halves of a pair (promp_device.h: pair_post / pair_wait); seq counts up within a segment
template <int H>
PROMP_DEV void pass2_sync(float* preg, int seq, int lane) {
    constexpr Pass2Lds Q = pass2_layout(0);
    pair_post(preg + Q.flags, H, seq, lane);
    pair_wait(preg + Q.flags, 1 - H, seq);
}
PROMP_DEV u32x4 pass2_read_chunk(const float* p) { return join_w2(*(const u32x2*)p, *(const u32x2*)(p + 32)); }

template <int H, bool BWD, bool STORE>
PROMP_DEV void pass2_tile(Pass2Sums& S, float (&xr)[8], const PassWalk& W, const PassTileAddr& T, float* sm, float* preg, int lane,
                          int t, bool valid, int tnext) {
    constexpr int NC1 = 4, NC2 = 4, H1 = 64, H2 = 64, NP1 = 2, NP2 = 2;
    constexpr int HCR = chain_cache_row(H1, H2);
    constexpr int TPL = PROMP_PASS_TPLANE, XPL = PROMP_PASS_XPLANE, DPL = PROMP_PASS_DPLANE;
    constexpr PassLds L = pass_layout(NC1, NC2, 1, 0);
    constexpr Pass2Lds Q = pass2_layout(0);
    constexpr int PS = L.n_frag, OT = 1 - H;
    const int i16 = lane & 15, kk = lane >> 4;
    const u32x4* F = (const u32x4*)(sm + L.wp) + lane;
    const bool own0 = 2 * kk < W.A, own1 = 2 * kk + 1 < W.A;
    const int q0 = own0 ? 2 * kk : 0, q1 = own1 ? 2 * kk + 1 : 0;
    float *XT = preg + Q.xt0 + H * Q.xt_stride, *TA = preg + Q.ta, *TB = preg + Q.tb, *MU = preg + Q.mu, *DM = preg + Q.dm;
    const int left = W.tnrows - 16 * t;
    const int nrows = valid ? (left < 16 ? left : 16) : 0;
    const long long base = (long long)W.trow0 + (valid ? 16 * t : 0);            // (a null tile reads the task's first rows, weight 0)
    const bool rvalid = i16 < nrows;
    const long long n = base + (rvalid ? i16 : 0);
    float* const hcb = STORE ? W.hcache + (base + 16 * W.task) * HCR + 16 * i16 + 4 * kk : nullptr;

    // ---- region 0: row data, observation planes (private tile), layer 1 of the own blocks
    PASS_STAMP(0);
    u32x4 w1f[3][2];
    pass_load_frags<2>(w1f, F, PS, L.f_w1 + 2 * H * 64, 64);
    const float* olsp = W.old_log_std + (W.ls_per_row ? n * W.A : (long long)W.task * W.A);
    const float advn = W.adv[n] * (rvalid ? 1.f : 0.f);
    const float ac0 = W.act[n * W.A + q0], ac1 = W.act[n * W.A + q1];
    const float mo0 = W.old_mean[n * W.A + q0], mo1 = W.old_mean[n * W.A + q1];
    const float so0 = olsp[q0], so1 = olsp[q1];
    u32x4 xB[3];
    {
        f32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = xr[e];
            hi[e] = xr[4 + e];
        }
        pass_split8(lo, hi, xB);
    }
    if (BWD) {
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            sts_w2(XT + tt * XPL + T.xw0, xB[tt][0], xB[tt][1]);
            sts_w2(XT + tt * XPL + T.xw1, xB[tt][2], xB[tt][3]);
        }
    }
    f32x4 h1[2], h2[2];
    {
        const float* B1l = sm + L.b1 + 4 * kk + 16 * (2 * H);
#pragma unroll
        for (int j = 0; j < 2; ++j) h1[j] = lds4(B1l + 16 * j);
    }
    pass_gemm16<2>(h1, w1f, xB);                                  // Z1^T (own blocks) = W1^T X^T + b1
    sched_fence();
    // ---- region 1: tanh, own hidden_0 planes (-> tile A, chunk H); layer 2 starts on the own chunk
    PASS_STAMP(1);
    u32x4 hBo[3];
    {
        u32x4 w2fo[3][2];
        pass_load_frags<2>(w2fo, F, PS, L.f_w2f + (2 * H * NP1 + H) * 64, NP1 * 64);       // [c2 = 2 H + j][P = H]
        {
            const float* B2l = sm + L.b2 + 4 * kk + 16 * (2 * H);
#pragma unroll
            for (int j = 0; j < 2; ++j) h2[j] = lds4(B2l + 16 * j);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            h1[j] = pass_tanh4(h1[j]);
            if (STORE && valid) *(f32x4*)(hcb + 256 * (2 * H + j)) = h1[j];
        }
        pass_split8(h1[0], h1[1], hBo);
        pass_store_planes(TA, TPL, T.wr + 256 * H, hBo);          // (over the own dZ1 planes of the previous tile: own reads only)
        pass_gemm16<2>(h2, w2fo, hBo);
    }
    {
        u32x4 w2fx[3][2], hBx[3];
        pass2_sync<H>(preg, 3 * W.tix + 1, lane);                 // B1: both halves' hidden_0 planes are in tile A
        PASS_STAMP(2);
        pass_load_frags<2>(w2fx, F, PS, L.f_w2f + (2 * H * NP1 + OT) * 64, NP1 * 64);  // [c2 = 2 H + j][P = 1 - H]
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) hBx[tt] = pass2_read_chunk(TA + tt * TPL + T.wr + 256 * OT);
        pass_gemm16<2>(h2, w2fx, hBx);                            // Z2^T (own blocks) = W2^T H1^T + b2
    }
    sched_fence();
    // ---- region 3: tanh, own hidden_1 planes (-> tile B, chunk H: read back by this wave only); the own half of
    //      mu^T = W3^T H2^T (+ b3 in half 0); the halves cross through LDS
    PASS_STAMP(3);
    float mu0, mu1;
    {
        u32x4 hB2o[3], w3f[3][1];
        pass_load_frags<1>(w3f, F, PS, L.f_w3f + H * 64, 0);     // own chunk of the output kernel
        f32x4 m[2] = {zero4(), zero4()};
        if (H == 0) {
            const f32x2 bb = lds2(sm + L.b3 + 2 * kk);
            m[0][0] = bb[0];
            m[0][1] = bb[1];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            h2[j] = pass_tanh4(h2[j]);
            if (STORE && valid) *(f32x4*)(hcb + 256 * (NC1 + 2 * H + j)) = h2[j];
        }
        pass_split8(h2[0], h2[1], hB2o);
        if (BWD) pass_store_planes(TB, TPL, T.wr + 256 * H, hB2o);   // (the partner read the own dZ2 planes of the previous tile before B1)
        PASS_STAMP(4);
#pragma unroll
        for (int ta = 2; ta >= 0; --ta)
#pragma unroll
            for (int tb = 2 - ta; tb >= 0; --tb) m[(ta + tb) & 1] = mfma16_bf16w(w3f[ta][0], hB2o[tb], m[(ta + tb) & 1]);
        f32x2 mine;
        mine[0] = m[0][0] + m[1][0];
        mine[1] = m[0][1] + m[1][1];
        *(f32x2*)(MU + 128 * H + 2 * lane) = mine;               // (the partner read the previous tile's halves before its B3)
        pass2_sync<H>(preg, 3 * W.tix + 2, lane);                 // B2
        const f32x2 other = *(const f32x2*)(MU + 128 * OT + 2 * lane);
        mu0 = mine[0] + other[0];
        mu1 = mine[1] + other[1];
        if (STORE && valid && H == 0) {
            f32x2 mm;
            mm[0] = mu0;
            mm[1] = mu1;
            *(f32x2*)(W.hcache + (base + 16 * W.task) * HCR + 256 * (NC1 + NC2) + 8 * i16 + 2 * kk) = mm;
        }
    }
    PASS_STAMP(5);
    float d0, d1;
    {   // lane (i16, kk) = sample i16, actions 2 kk and 2 kk + 1 (both halves: the epilogue is per row, cheaper than a third exchange)
        const f32x4 dc0 = lds4(sm + Q.dtab + 8 * lane), dc1 = lds4(sm + Q.dtab + 8 * lane + 4);
        const float Ws0 = dc0[0], Ws1 = dc0[1], We0 = dc0[2], We1 = dc0[3], Wsn20 = dc1[0], Wsn21 = dc1[1], Wrden0 = dc1[2], Wrden1 = dc1[3];
        const float o0 = own0 ? 1.f : 0.f, o1 = own1 ? 1.f : 0.f, rv = rvalid ? 1.f : 0.f;
        const float z0 = (ac0 - mu0) * We0, z1 = (ac1 - mu1) * We1;
        const float zo0 = (ac0 - mo0) * fast_exp(-so0), zo1 = (ac1 - mo1) * fast_exp(-so1);
        const float num0 = (mo0 - mu0) * (mo0 - mu0) + fast_exp(2.f * so0) - Wsn20;
        const float num1 = (mo1 - mu1) * (mo1 - mu1) + fast_exp(2.f * so1) - Wsn21;
        float dlp = o0 * ((so0 - Ws0) - 0.5f * (z0 * z0 - zo0 * zo0)) + o1 * ((so1 - Ws1) - 0.5f * (z1 * z1 - zo1 * zo1));
        float sumz2 = o0 * (z0 * z0) + o1 * (z1 * z1);
        float kl = o0 * (num0 * Wrden0 + Ws0 - so0) + o1 * (num1 * Wrden1 + Ws1 - so1);
        dlp = fold_groups16(dlp);          // sums over the row's actions (the four lane groups)
        sumz2 = fold_groups16(sumz2);
        kl = fold_groups16(kl);
        const float rho = expf(rvalid ? dlp : 0.f);        // (padding rows: a finite ratio with zero weight)
        const float aw = advn * W.invN;
        const float x = rho * advn, y = fminf(fmaxf(rho, 1.f - W.clip_eps), 1.f + W.clip_eps) * advn;
        const float lp = -W.sums - 0.5f * sumz2 - 0.5f * (float)W.A * 1.8378770664093453f;
        // d loss / d logpi (c: KL 0, ratio -aw rho, clip -aw rho where the unclipped term is the smaller one, log-likelihood -aw),
        // the weight of the KL cotangents (ck; KL only)
        const float c_r = -aw * rho, c_c = (x <= y) ? c_r : 0.f;
        const float c = pass2_pick(W, 0.f, c_r, c_c, -aw);
        const float ck = pass2_pick(W, rv * W.invN, 0.f, 0.f, 0.f);
        const float dklm0 = -2.f * (mo0 - mu0) * Wrden0, dklm1 = -2.f * (mo1 - mu1) * Wrden1;
        d0 = o0 * (c * z0 * We0 + ck * dklm0);
        d1 = o1 * (c * z1 * We1 + ck * dklm1);
        if (H == 0) {                      // the row's scalars and the distribution's gradients are summed by half 0
            const float den0 = 2.f * Wsn20 + 1e-8f, den1 = 2.f * Wsn21 + 1e-8f;
            const float lrow = pass2_pick(W, kl * W.invN, -rho * aw, -fminf(x, y) * W.invN, -lp * aw);
            const float first = (kk == 0) ? rv : 0.f;      // one lane per row carries the row's scalars
            S.loss += first * lrow;
            S.klsum += first * (kl * W.invN);
            const float dkls0 = (-2.f * Wsn20 * den0 - 4.f * num0 * Wsn20) * (Wrden0 * Wrden0) + 1.f;
            const float dkls1 = (-2.f * Wsn21 * den1 - 4.f * num1 * Wsn21) * (Wrden1 * Wrden1) + 1.f;
            S.gs0 += o0 * (c * (z0 * z0 - 1.f) + ck * dkls0);
            S.gs1 += o1 * (c * (z1 * z1 - 1.f) + ck * dkls1);
            S.gb30 += d0;
            S.gb31 += d1;
        }
    }
    if (!BWD) {
        pass_load_x(xr, W, tnext, i16, kk);
        return;
    }
    // ---- region 6: the mean cotangents' planes (-> the pair's tile: both halves write the same values); dH2^T (own blocks) = W3 dmu^T
    //      on the exact FP32 instruction (K = act_dim); output-kernel gradient of the own units
    PASS_STAMP(6);
    f32x4 dz2[2];
    {
        // (element by element into scalars first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whichever
        //  element is named -- hipcc 7.2, seen in the listing and on the device)
        const f32x4 ct0 = lds4(sm + Q.ctab + 8 * lane);
        const float cf0 = ct0[0], cf1 = ct0[1], cf2 = ct0[2], cf3 = ct0[3], cf4 = sm[Q.ctab + 8 * lane + 4];
        const int Trd16_0 = __builtin_bit_cast(int, cf0), Trd16_1 = __builtin_bit_cast(int, cf1);
        const int Tdr0 = __builtin_bit_cast(int, cf2), Tdr1 = __builtin_bit_cast(int, cf3), Tdmw = __builtin_bit_cast(int, cf4);
        unsigned dw[3];
        bf16_split3_pair(d0, d1, dw);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) DM[tt * DPL + Tdmw] = __builtin_bit_cast(float, dw[tt]);
        u32x4 bD[3], aH[2][3];
        pass_read_tr(bD, DM, DPL, Tdr0, Tdr1);
#pragma unroll
        for (int j = 0; j < 2; ++j) pass_read_tr(aH[j], TB, TPL, Trd16_0 + 2 * (128 * H + 16 * j), Trd16_1 + 2 * (128 * H + 16 * j));
        f32x2 wb[2];
        {
            const float* W3b = sm + L.w3b + lane * 2 + 2 * H * 128;
#pragma unroll
            for (int j = 0; j < 2; ++j) wb[j] = lds2(W3b + j * 128);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) dz2[j] = mfma16(wb[j][0], d0, zero4());
#pragma unroll
        for (int j = 0; j < 2; ++j) dz2[j] = mfma16(wb[j][1], d1, dz2[j]);
#pragma unroll
        for (int ta = 2; ta >= 0; --ta)
#pragma unroll
            for (int tb = 2 - ta; tb >= 0; --tb)
#pragma unroll
                for (int j = 0; j < 2; ++j) S.aw3[j] = mfma16_bf16w(aH[j][ta], bD[tb], S.aw3[j]);
    }
    sched_fence();
    // ---- region 7: dZ2, its planes (-> tile B, chunk H); hidden_1 kernel gradient aw2[:, own] = H1^T dZ2[own] (both halves' hidden_0
    //      planes: read before B3, while the partner's are still there); dH1 starts on the own chunk
    PASS_STAMP(7);
    f32x4 ad1[2] = {zero4(), zero4()};
    {
        u32x4 dB2o[3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dz2[j][r] *= -pass_neg_dtanh(h2[j][r]);
                S.gb2[j][r] += dz2[j][r];
            }
        pass_split8(dz2[0], dz2[1], dB2o);
        pass_store_planes(TB, TPL, T.wr + 256 * H, dB2o);        // (after this wave's reads of its hidden_1 planes: a memory dependence)
        {
            u32x4 fa[2][3], fb[3];
#pragma unroll
            for (int b = 0; b < 2; ++b) pass_read_tr(fa[b], TA, TPL, T.rd32_0 + 256 * b, T.rd32_1 + 256 * b);
            pass_read_tr(fb, TB, TPL, T.rd32_0 + 256 * H, T.rd32_1 + 256 * H);
            u32x4 w2bo[3][2];
            pass_load_frags<2>(w2bo, F, PS, L.f_w2b + (2 * H * NP2 + H) * 64, NP2 * 64);   // [c1 = 2 H + j][P = H]
            pass_gemm16<2>(ad1, w2bo, dB2o);
#pragma unroll
            for (int ta = 2; ta >= 0; --ta)
#pragma unroll
                for (int tb = 2 - ta; tb >= 0; --tb)
#pragma unroll
                    for (int bi = 0; bi < 2; ++bi) S.aw2[bi] = mfma32_bf16w(fa[bi][ta], fb[tb], S.aw2[bi]);
        }
    }
    {
        u32x4 w2bx[3][2], dB2x[3];
        pass2_sync<H>(preg, 3 * W.tix + 3, lane);                 // B3: both halves' dZ2 planes are in tile B; all reads of the H1 planes done
        PASS_STAMP(8);
        pass_load_x(xr, W, tnext, i16, kk);                       // the next tile's observations (a tile's worth of latency ahead)
        pass_load_frags<2>(w2bx, F, PS, L.f_w2b + (2 * H * NP2 + OT) * 64, NP2 * 64);      // [c1 = 2 H + j][P = 1 - H]
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) dB2x[tt] = pass2_read_chunk(TB + tt * TPL + T.wr + 256 * OT);
        pass_gemm16<2>(ad1, w2bx, dB2x);                          // dH1^T (own blocks) = W2 dZ2^T
    }
    if (STORE && valid) {
#pragma unroll
        for (int j = 0; j < 2; ++j) *(f32x4*)(hcb + 256 * (NC1 + NC2) + 128 + 256 * (2 * H + j)) = ad1[j];
    }
    sched_fence();
    // ---- region 9: dZ1, its planes (-> tile A, chunk H: over the own hidden_0 planes), hidden_0 kernel gradient aw1[:, own] = X^T dZ1[own]
    PASS_STAMP(9);
    {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ad1[j][r] *= -pass_neg_dtanh(h1[j][r]);
                S.gb1[j][r] += ad1[j][r];
            }
        u32x4 dB1o[3], fd[3], fx[3];
        pass_read_tr(fx, XT, XPL, T.rd32_0, T.rd32_1);
        pass_split8(ad1[0], ad1[1], dB1o);
        pass_store_planes(TA, TPL, T.wr + 256 * H, dB1o);
        pass_read_tr(fd, TA, TPL, T.rd32_0 + 256 * H, T.rd32_1 + 256 * H);
#pragma unroll
        for (int ta = 2; ta >= 0; --ta)
#pragma unroll
            for (int tb = 2 - ta; tb >= 0; --tb) S.aw1 = mfma32_bf16w(fx[ta], fd[tb], S.aw1);
    }
    sched_fence();
    PASS_STAMP(10);
}

// Cross-pair, fixed-order sum of the pairs' gradient tiles -> one partial row in global memory.  The two halves of a pair fill
// the pair's LDS slab of [NP + 2] floats between them (every entry written exactly once), then all threads add the slabs in pair
// order.
template <int H>
PROMP_DEV void pass2_store_slab(float* mine, const Pass2Sums& S, float gs0, float gs1, float gb30, float gb31, float loss,
                                float klsum, int O, int A, int lane) {
    constexpr int H1 = 64, H2 = 64;
    // (the opaque zero keeps the ~80 store addresses from being hoisted out of the segment loop -- they would be spilled to
    //  scratch at the head of the kernel and reloaded here, one round trip each)
    lane += opaque_zero();
    const int i16 = lane & 15, kk = lane >> 4, j32 = lane & 31, kh = lane >> 5;
    const int ob1 = O * H1, oW2 = ob1 + H1, ob2 = oW2 + H1 * H2, oW3 = ob2 + H2, ob3 = oW3 + H2 * A, oS = ob3 + A, NP = oS + A;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[oW2 + (32 * bi + (r & 3) + 8 * (r >> 2) + 4 * kh) * H2 + 32 * H + j32] = S.aw2[bi][r];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;      // observation index
        if (row < O) mine[row * H1 + 32 * H + j32] = S.aw1[r];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (i16 < A) mine[oW3 + (16 * (2 * H + j) + 4 * kk + r) * A + i16] = S.aw3[j][r];
    if (i16 == 0) {           // (bias sums already folded over the 16 sample lanes)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mine[ob1 + 16 * (2 * H + j) + 4 * kk + r] = S.gb1[j][r];
                mine[ob2 + 16 * (2 * H + j) + 4 * kk + r] = S.gb2[j][r];
            }
        if (H == 0) {
            if (2 * kk < A) {
                mine[ob3 + 2 * kk] = gb30;
                mine[oS + 2 * kk] = gs0;
            }
            if (2 * kk + 1 < A) {
                mine[ob3 + 2 * kk + 1] = gb31;
                mine[oS + 2 * kk + 1] = gs1;
            }
        }
    }
    if (H == 0 && lane == 0) {
        mine[NP] = loss;
        mine[NP + 1] = klsum;
    }
}

template <int H, bool BWD, bool STORE>
PROMP_DEV void pass2_walk(Pass2Sums& S, PassWalk& W, const PassTileAddr& T, float* sm, float* preg, int lane, int pair, int tile0,
                          int rounds, float (&xr)[8]) {
    W.tix = 0;
    int t = tile0 + pair;
    for (int r = 0; r < rounds; ++r, t += PROMP_PASS2_PAIRS) {
        pass2_tile<H, BWD, STORE>(S, xr, W, T, sm, preg, lane, t, t < W.tend, t + PROMP_PASS2_PAIRS);
        W.tix += 1;
    }
}

// Eight waves = four pairs; a workgroup's waves go to the SIMDs cyclically (MI355X_MICROARCH.md, LDS).
template <bool BWD, bool STORE>
__global__ void __launch_bounds__(512, 2) k_pass_pair(PassArgs a) {
    constexpr int H1 = 64, H2 = 64, NPAIR = PROMP_PASS2_PAIRS;
    constexpr int DPL = PROMP_PASS_DPLANE;
    constexpr PassLds L = pass_layout(4, 4, 1, 0);
    constexpr Pass2Lds Q = pass2_layout(0);
    PROMP_SMEM_DECL;
    float* sm = (float*)PROMP_SMEM_PTR;
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
#ifndef PROMP_PAIR_MAP
#define PROMP_PAIR_MAP 1
#endif
    // (waves w and w + 4 of a workgroup share a SIMD: with pair = w / 2 every SIMD hosts waves of two different pairs, which drift
    //  apart -- one in a matrix phase while the other splits; PROMP_PAIR_MAP 0 puts the two halves of a pair on one SIMD)
    const int pair = PROMP_PAIR_MAP ? (w >> 1) : (w & (NPAIR - 1)), half = PROMP_PAIR_MAP ? (w & 1) : (w >> 2);
    const int i16 = lane & 15, kk = lane >> 4;
    const int O = a.O, A = a.A;
    const int ob1 = O * H1, oW2 = ob1 + H1, ob2 = oW2 + H1 * H2, oW3 = ob2 + H2, ob3 = oW3 + H2 * A, oS = ob3 + A, NP = oS + A;
    const float* dist = sm + L.dist;
    float* preg = sm + Q.pair0 + pair * Q.pair_stride;
    PassTileAddr T;
    {
        const int p16 = lane & 15, g32 = (lane >> 4) & 1, kh = lane >> 5;
        T.wr = 2 * pass_slot(i16, kk);
        T.xw0 = 2 * pass_slot(i16, 2 * kk);
        T.xw1 = 2 * pass_slot(i16, 2 * kk + 1);
        T.rd32_0 = 2 * pass_slot(8 * kh + (p16 >> 2), 4 * g32 + (p16 & 3));
        T.rd32_1 = 2 * pass_slot(8 * kh + 4 + (p16 >> 2), 4 * g32 + (p16 & 3));
        T.rd16_0 = T.rd16_1 = T.dmw = T.dr0 = T.dr1 = 0;        // (from the table below)
        if (w == 0) {     // the addresses of the cotangent-tile accesses (pass_tile in promp_kernels_pass.h keeps them in registers)
            // (stored the way the tiles load them -- one four-float vector, one float: a store through another type, scalar floats
            //  included, is a store the compiler may assume the vector load never sees; it deleted them)
            f32x4 c4;
            c4[0] = __builtin_bit_cast(float, 2 * pass_slot(8 * (kk & 1) + (p16 >> 2), p16 & 3));
            c4[1] = __builtin_bit_cast(float, 2 * pass_slot(8 * (kk & 1) + 4 + (p16 >> 2), p16 & 3));
            c4[2] = __builtin_bit_cast(float, (kk < 2) ? 8 * (8 * kk + (p16 >> 2)) + 32 * kk + 2 * (p16 & 3) : 4);
            c4[3] = __builtin_bit_cast(float, (kk < 2) ? 8 * (8 * kk + 4 + (p16 >> 2)) + 32 * kk + 2 * (p16 & 3) : 4);
            sts4(sm + Q.ctab + 8 * lane, c4);
            sm[Q.ctab + 8 * lane + 4] = __builtin_bit_cast(float, 8 * i16 + 32 * (i16 >> 3) + kk);
        }
    }
    PassWalk W;
    W.obs = a.obs; W.act = a.act; W.adv = a.adv; W.old_mean = a.old_mean; W.old_log_std = a.old_log_std; W.hcache = a.hcache;
    W.ls_per_row = a.ls_per_row; W.O = O; W.A = A; W.loss_kind = a.loss_kind; W.clip_eps = a.clip_eps;
    // The objective kind as bit masks in vector registers, and the epilogue's four-way choices as AND / OR of float bits: hipcc 7.2
    // turns the select chain (KL / ratio / clip / log-likelihood) into branches -- scalar or exec-masked, whichever file the kind
    // lives in -- and in the half-1 instance of this kernel the log-likelihood arm's assignment is lost (its weight arrived
    // undefined; found on the device, tools/pair_debug.py, and visible in the listing).  Bit masks leave no control flow to build.
    {
        const int lk = wave_uniform(a.loss_kind);
        unsigned mr = lk == LOSS_RATIO ? ~0u : 0u, mc = lk == LOSS_CLIP ? ~0u : 0u, ml = lk == LOSS_LOGLIK ? ~0u : 0u, mk = lk == LOSS_KL ? ~0u : 0u;
        mr = (unsigned)wave_uniform((int)mr); mc = (unsigned)wave_uniform((int)mc); ml = (unsigned)wave_uniform((int)ml); mk = (unsigned)wave_uniform((int)mk);
        pin_s(mr); pin_s(mc); pin_s(ml); pin_s(mk);       // (opaque, but scalar registers)
        W.m_ratio = mr; W.m_clip = mc; W.m_ll = ml; W.m_kl = mk;
    }
    W.dbg = a.dbg; W.tix = 0;
    W.own0 = 2 * kk < A; W.own1 = 2 * kk + 1 < A;
    W.q0 = W.own0 ? 2 * kk : 0; W.q1 = W.own1 ? 2 * kk + 1 : 0;
    W.s0 = W.s1 = W.e0 = W.e1 = W.sn20 = W.sn21 = W.rden0 = W.rden1 = 0.f;      // (read from the table once per tile)

#ifdef PROMP_PAIR_PRIO
    if (w >= 4) wave_priority(1);          // developer switch: the later-placed half of the waves loses every issue arbitration otherwise
#endif
    const int sg0 = a.wg_seg_offsets[blockIdx.x], sg1 = a.wg_seg_offsets[blockIdx.x + 1];
    CH_WGSTAMP(0);
    for (int sg = sg0; sg < sg1; ++sg) {
        const ChainSeg seg = a.segs[sg];
        W.task = seg.task;
        W.trow0 = a.task_row_offsets[seg.task];
        W.tnrows = a.task_row_offsets[seg.task + 1] - W.trow0;
        W.invN = 1.0f / (float)W.tnrows;
        W.tend = seg.tile0 + seg.ntiles;
        const int rounds = (seg.ntiles + NPAIR - 1) / NPAIR;
        const float* th = a.theta + (long long)seg.task * a.theta_task_stride;
        // the first tile's observations are on their way while the network is staged
        float xr[8];
        pass_load_x(xr, W, seg.tile0 + pair, i16, kk);
        const ChainDistRaw draw = chain_dist_load(th, nullptr, oS, A, tid);
        __syncthreads();
        CH_STAMP(0);
        pass_stage_net<4, 4, 8>(sm, th, O, A, tid, []() {});
        chain_stage_dist(sm + L.dist, draw, A, a.clip_log_std, a.min_log_std, tid);
        // the action slots >= 8 of the cotangent tiles read as zero (the end-of-segment slabs alias them: once per segment)
        if (half == 0) {
            for (int e = lane; e < 3 * DPL; e += 64) preg[Q.dm + e] = 0.f;
            if (lane < 4) preg[Q.flags + lane] = 0.f;
        }
        __syncthreads();
        CH_STAMP(1);
        if (w == 0) {     // the distribution parameters of every lane's two actions (read back by all waves once per tile)
            f32x4 d0, d1;
            d0[0] = dist[CH_LS + W.q0]; d0[1] = dist[CH_LS + W.q1]; d0[2] = dist[CH_ES + W.q0]; d0[3] = dist[CH_ES + W.q1];
            d1[0] = dist[CH_SN2 + W.q0]; d1[1] = dist[CH_SN2 + W.q1]; d1[2] = dist[CH_RDEN + W.q0]; d1[3] = dist[CH_RDEN + W.q1];
            sts4(sm + Q.dtab + 8 * lane, d0);
            sts4(sm + Q.dtab + 8 * lane + 4, d1);
        }
        __syncthreads();
        W.sums = 0.f;                                 // sum of the log standard deviations (log-likelihood objective)
        for (int aa = 0; aa < A; ++aa) W.sums += dist[CH_LS + aa];

        Pass2Sums S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S.aw2[0][r] = S.aw2[1][r] = S.aw1[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            S.aw3[j] = zero4();
            S.gb1[j] = zero4();
            S.gb2[j] = zero4();
        }
        S.loss = S.klsum = S.gs0 = S.gs1 = S.gb30 = S.gb31 = 0.f;

#ifdef PROMP_PAIR_SKEW
        if (pair >= 2) __builtin_amdgcn_s_sleep(PROMP_PAIR_SKEW);      // developer switch: start half the pairs late
#endif
        if (half == 0) pass2_walk<0, BWD, STORE>(S, W, T, sm, preg, lane, pair, seg.tile0, rounds, xr);
        else pass2_walk<1, BWD, STORE>(S, W, T, sm, preg, lane, pair, seg.tile0, rounds, xr);
        CH_STAMP(2);

        float* P = a.partials + (long long)sg * a.partial_stride;
        float loss = S.loss, klsum = S.klsum, gs0 = S.gs0, gs1 = S.gs1, gb30 = S.gb30, gb31 = S.gb31;
        // per-action sums: lanes of one kk group differ in the sample; scalars live in the kk = 0 lanes
        gs0 = row16_sum(gs0);  gs1 = row16_sum(gs1);  gb30 = row16_sum(gb30);
        gb31 = row16_sum(gb31);  loss = row16_sum(loss);  klsum = row16_sum(klsum);
        if (!BWD) {   // only the two scalars leave the workgroup (half 0 of every pair holds them)
            lds_barrier();
            if (lane == 0 && half == 0) {
                sm[4 + 2 * pair] = loss;
                sm[4 + 2 * pair + 1] = klsum;
            }
            lds_barrier();
            if (tid == 0) {
                float l = 0.f, k = 0.f;
                for (int ww = 0; ww < NPAIR; ++ww) {
                    l += sm[4 + 2 * ww];
                    k += sm[4 + 2 * ww + 1];
                }
                P[NP] = l;
                P[NP + 1] = k;
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                S.gb1[j][r] = row16_sum(S.gb1[j][r]);
                S.gb2[j][r] = row16_sum(S.gb2[j][r]);
            }
        gs0 *= dist[CH_LMASK + W.q0];
        gs1 *= dist[CH_LMASK + W.q1];
        const int SL = (NP + 2 + 3) & ~3;
        CH_STAMP(5);
        lds_barrier();                // every wave is done with the parameter planes / transposed tiles
        CH_STAMP(6);
        if (half == 0) pass2_store_slab<0>(sm + 4 + pair * SL, S, gs0, gs1, gb30, gb31, loss, klsum, O, A, lane);
        else pass2_store_slab<1>(sm + 4 + pair * SL, S, gs0, gs1, gb30, gb31, loss, klsum, O, A, lane);
        lds_barrier();
        CH_STAMP(7);
#pragma unroll 2
        for (int e = 4 * tid; e < SL; e += 4 * 512) {
            f32x4 v[NPAIR];
#pragma unroll
            for (int ww = 0; ww < NPAIR; ++ww) v[ww] = *(const f32x4*)(sm + 4 + ww * SL + e);      // all slab reads in flight together
            f32x4 tsum = v[0];
#pragma unroll
            for (int ww = 1; ww < NPAIR; ++ww) tsum += v[ww];
            *(f32x4*)(P + e) = tsum;
        }
        CH_STAMP(3);
        CH_WGSTAMP(1 + (sg - sg0 < 2 ? sg - sg0 : 1));
    }
}
