// promp_kernels_pass.h -- objective + mean KL (+ gradient) of the tasks' slabs for hidden widths from {32, 64}, obs_dim <= 32
// (reference rows a8-a11):
//
//   k_pass : objective (ratio / PPO-clip / log-lik / KL), mean KL and gradient on one step's slabs                 (K8-K11)
//
// Arithmetic follows oracle/promp.py (which restates meta_algos/pro_mp.py:59-155, meta_algos/base.py:192-215,
// policies/networks/mlp.py:65-119, policies/distributions/diagonal_gaussian.py:16-109 of the reference).
//
// Design.  FP32 is a vector-rate format on gfx950 (v_mfma_f32_16x16x4_f32 = 64 FLOP / clock / SIMD); the 16-bit matrix pipe is
// sixteen times faster.  Every GEMM of this kernel therefore runs on that pipe in float32-EQUIVALENT arithmetic: both operands
// are split into terms of a 16-bit format and the significant cross products are accumulated in float32.  Since round 6: TWO FP16
// terms, THREE products (hi.hi + hi.lo + lo.hi; promp_device.h: split_pair) -- as accurate as the FP32 fma
// chain on the network's shapes as long as every split operand sits near 1, which the kernel arranges with exact powers of two
// that follow the data (obs_shift, pass_cotangent_scale below).  Rounds 3-5 (-DPROMP_SPLIT_TERMS=3): three BF16 terms, six of the
// nine products, no range to mind, twice the matrix instructions and 2.7 times the split instructions.  (Comments below that say
// "BF16 planes" describe the layout, which is the same for both: 16-bit halves in 32-bit words.)
//
//  * Register chain (as k_chain_hvp): a wave owns 16-sample tiles and keeps its activations TRANSPOSED, samples along n.
//    Lane (i16 = sample, kk) receives units 16 c + 4 kk + r of a layer as its four D registers; the eight k-slots a lane
//    feeds to v_mfma_f32_16x16x32_bf16 for the input chunk P are its own registers of the blocks 2P and 2P + 1, so a layer's
//    output becomes the next layer's B operand after a split in registers -- forward and backward alike.  The weights are
//    split ONCE per segment into BF16 planes in LDS, in fragment order for both orientations (one ds_read_b128 per MFMA
//    operand, lane-linear: conflict free).
//  * Weight gradients contract over samples and need both operands with the unit along the lane index.  Each operand's BF16
//    planes go through a per-wave LDS tile of 8-byte chunks (4 units of one sample) and come back through
//    ds_read_b64_tr_b16, the 4 x 16 transpose read of gfx950: two reads give a lane the eight samples of its unit.  The
//    products run on v_mfma_f32_32x32x16_bf16 (K = 16 samples: full utilisation with 16-sample tiles).  The chunk placement
//    pass_slot() is conflict free for the chain-side ds_write_b64 and for the transpose reads.
//  * The split terms are computed once per activation and feed both the chain GEMM and the weight-gradient GEMM.
//  * Only dH2 = W3 dmu^T (K = act_dim <= 8) stays on the FP32 instruction: a K = 32 instruction would be 3/4 padding.
//
// Work split: the segment table of k_chain_hvp (16-sample tiles, rounds of NW tiles, cost-balanced shares, one workgroup of
// NW = 4 waves -- one per SIMD, 512 registers -- per CU).  Every (workgroup, segment) writes one partial row; k_reduce_task
// adds a task's rows in slot order (bitwise reproducible).
#pragma once
#include "promp_device.h"
#include "promp_kernels_chain.h"

// The six products (A-side term ta, B-side term tb) with ta + tb <= 2 are walked by A-side term: (2,0) (1,1) (1,0) (0,2) (0,1) (0,0)
// -- roughly smallest first, and an A-side fragment is read once and feeds one to three consecutive instructions.
// (1,2), (2,1), (2,2) are below 2^-24 of the result and dropped.

struct PassLds {                 // offsets in 4-byte words
    // BF16 planes of the segment's network: [term 3][fragment NG][4 words]; a fragment = one lane's operand of one MFMA.
    // Fragment order: W1 [c][lane] | W2 forward [c2][P][lane] | W2 backward [c1][P][lane] | W3 forward [P][lane]
    int wp, plane_stride;
    int f_w1, f_w2f, f_w2b, f_w3f, n_frag;      // first fragment of each region, fragment count
    int w3b, b1, b2, b3, dist, n_side;          // float32 side tables (contiguous from w3b: W3 for dH2, biases), their count
    int wave0, wave_stride, xt, ta, tb, dm;
    int total;
};
// (PROMP_PASS_TPLANE / XPLANE / DPLANE, PassTileAddr: promp_kernels_chain.h -- k_chain_hvp's cached instance uses the same tiles)

PROMP_CX PassLds pass_layout(int NC1, int NC2, int nwaves, int NP) {
    PassLds L{};
    int o = 4;                                            // [0, 4): spare (the end-of-segment slabs start at 4)
    L.f_w1 = 0;                                           // W1[obs 8 kk + e][16 c + i16] (x tanh prescale)
    L.f_w2f = L.f_w1 + NC1 * 64;                          // W2[u(P, kk, e)][16 c2 + i16] (x tanh prescale)
    L.f_w2b = L.f_w2f + NC2 * (NC1 / 2) * 64;             // W2[16 c1 + i16][u(P, kk, e)]
    L.f_w3f = L.f_w2b + NC1 * (NC2 / 2) * 64;             // W3[u(P, kk, e)][action of row i16]
    L.n_frag = L.f_w3f + (NC2 / 2) * 64;
    L.plane_stride = 4 * L.n_frag;
    L.wp = o; o += PROMP_NT * L.plane_stride;
    L.w3b = o; o += NC2 * 128;                            // [c][lane][ro]: W3[16 c + i16][2 kk + ro]
    L.b1 = o; o += 16 * NC1;
    L.b2 = o; o += 16 * NC2;
    L.b3 = o; o += 8;
    L.n_side = o - L.w3b;
    L.dist = o; o += 48;
    L.wave0 = o;
    int q = 0;
    L.xt = q; q += PROMP_NT * PROMP_PASS_XPLANE;
    L.ta = q; q += PROMP_NT * PROMP_PASS_TPLANE;
    L.tb = q; q += PROMP_NT * PROMP_PASS_TPLANE;
    L.dm = q; q += PROMP_NT * PROMP_PASS_DPLANE;
    L.wave_stride = q;
    o += nwaves * q;
    {   // end of segment: one slab of [NP + 2] floats per wave, from offset 4 (aliases everything else)
        const int need = 4 + nwaves * ((NP + 2 + 3) & ~3);
        if (o < need) o = need;
    }
    L.total = o;
    return L;
}

// unit a lane's k-slot e of input chunk P stands for: its own D register (block 2P + e / 4, row e % 4)
PROMP_CX int pass_unit(int P, int kk, int e) { return 16 * (2 * P + (e >> 2)) + 4 * kk + (e & 3); }

// tanh of a pre-activation that arrives scaled by PROMP_TANH_PRESCALE, and h^2 - 1 (the NEGATED derivative): unpacked float32
// instructions (packed-f32 VALU beside MFMAs costs more than the issue slot it saves: MI355X_MICROARCH.md)
PROMP_DEV float pass_tanh(float y) { return __builtin_fmaf(fast_rcp(fast_exp2(y) + 1.f), -2.f, 1.f); }
// four at a time, stage by stage: a transcendental's result is not consumed by the very next instruction (no hazard s_nop)
PROMP_DEV f32x4 pass_tanh4(f32x4 y) {
    f32x4 e, r, h;
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = fast_exp2(y[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] += 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fast_rcp(e[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = __builtin_fmaf(r[i], -2.f, 1.f);
    return h;
}
PROMP_DEV float pass_neg_dtanh(float h) { return __builtin_fmaf(h, h, -1.f); }

// The segment's network -> BF16 planes (both orientations of the hidden_1 kernel) and float32 side tables in LDS.  A wave
// stages whole blocks of 64 fragments (one MFMA operand of every lane): the block is wave-uniform, so a fragment's eight source
// indices are a lane-constant pattern (pass_unit / the observation slots) scaled and shifted by scalars -- no per-element index
// arithmetic.  All global loads of a wave are issued before its first split (one round trip to L2); padding reads element 0 and
// is multiplied by 0 (a select would come back as an exec-masked branch around the load).
// `mid` runs between the load phase and the first split (all loads issued, none used, LDS untouched): the caller requests its first
// tile's observations there and joins the workgroup (see chain_stage_nets).
template <int NC1, int NC2, int NW, typename Mid>
PROMP_DEV void pass_stage_net(float* sm, const float* th, int O, int A, int tid, float w1s, Mid&& mid) {
    constexpr int H1 = 16 * NC1, H2 = 16 * NC2, NP1 = NC1 / 2, NP2 = NC2 / 2, NT = 64 * NW;
    constexpr PassLds L = pass_layout(NC1, NC2, 1, 0);
    constexpr int B1 = L.f_w2f / 64, B2 = L.f_w2b / 64, B3 = L.f_w3f / 64, NBLK = L.n_frag / 64;
    constexpr int IS = (L.n_side + NT - 1) / NT;
    const int lane = tid & 63, i16 = lane & 15, kk = lane >> 4, w = wave_uniform(tid >> 6);
    const int ob1 = O * H1, oW2 = ob1 + H1, ob2 = oW2 + H1 * H2, oW3 = ob2 + H2, ob3 = oW3 + H2 * A;
    // output row m = i16 of the mean GEMM carries action 2 (m / 4) + m % 4 for m % 4 < 2 (a lane's D registers 0, 1 are then
    // actions 2 kk, 2 kk + 1: the distribution epilogue keeps all four lane groups busy), nothing otherwise
    const int aro = i16 & 3, aact = 2 * (i16 >> 2) + aro;
    const bool aok = aro < 2 && aact < A;
    // One loop per block kind with a compile-time trip count; a wave whose block index runs past the end of a kind loads the
    // kind's last block again and skips the store.  The load phase neither branches nor uses a loaded value: all loads of the
    // wave are in flight together (with a branch per block kind the compiler waits, at the head of the next branch, for loads it
    // believes may still target the registers it reuses, and the round trips to L2 add up).
    constexpr int NK1 = B1, NK2 = B2 - B1, NK3 = B3 - B2, NK4 = NBLK - B3;
    constexpr int IT1 = (NK1 + NW - 1) / NW, IT2 = (NK2 + NW - 1) / NW, IT3 = (NK3 + NW - 1) / NW, IT4 = (NK4 + NW - 1) / NW;
    float x1[IT1][8], x2[IT2][8], x3[IT3][8], x4[IT4][8], y[IS];
#pragma unroll
    for (int it = 0; it < IT1; ++it) {              // W1[obs 8 kk + e][16 b + i16]
        const int bj = w + it * NW, b = bj < NK1 ? bj : NK1 - 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int o = 8 * kk + e;
            x1[it][e] = th[(o < O ? o : 0) * H1 + 16 * b + i16];
        }
    }
#pragma unroll
    for (int it = 0; it < IT2; ++it) {              // W2[u(P, kk, e)][16 c2 + i16]
        const int bj = w + it * NW, b = bj < NK2 ? bj : NK2 - 1, c2 = b / NP1, P = b - c2 * NP1;
#pragma unroll
        for (int e = 0; e < 8; ++e) x2[it][e] = th[oW2 + pass_unit(0, kk, e) * H2 + (32 * P * H2 + 16 * c2) + i16];
    }
#pragma unroll
    for (int it = 0; it < IT3; ++it) {              // W2[16 c1 + i16][u(P, kk, e)]
        const int bj = w + it * NW, b = bj < NK3 ? bj : NK3 - 1, c1 = b / NP2, P = b - c1 * NP2;
#pragma unroll
        for (int e = 0; e < 8; ++e) x3[it][e] = th[oW2 + i16 * H2 + pass_unit(0, kk, e) + (16 * c1 * H2 + 32 * P)];
    }
#pragma unroll
    for (int it = 0; it < IT4; ++it) {              // W3[u(P, kk, e)][action of row i16]
        const int bj = w + it * NW, P = bj < NK4 ? bj : NK4 - 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) x4[it][e] = th[oW3 + (pass_unit(0, kk, e) + 32 * P) * A + (aok ? aact : 0)];
    }
    // float32 side tables: W3 for dH2 = W3 dmu^T ([c][lane][ro]: W3[16 c + i16][2 kk + ro]), biases (those that feed a tanh prescaled)
#pragma unroll
    for (int it = 0; it < IS; ++it) {
        const int ej = tid + it * NT, e = ej < L.n_side ? ej : L.n_side - 1, wd = L.w3b + e;      // (past the end: the last entry again)
        int idx = 0;
        float m = 0.f;
        if (wd < L.b1) {
            const int c = e >> 7, l = (e >> 1) & 63, aa = 2 * (l >> 4) + (e & 1);
            idx = oW3 + (16 * c + (l & 15)) * A + (aa < A ? aa : 0);
            m = aa < A ? 1.f : 0.f;
        } else if (wd < L.b2) {
            idx = ob1 + (wd - L.b1);
            m = PROMP_TANH_PRESCALE;
        } else if (wd < L.b3) {
            idx = ob2 + (wd - L.b2);
            m = PROMP_TANH_PRESCALE;
        } else {
            const int aa = wd - L.b3;
            idx = ob3 + (aa < A ? aa : 0);
            m = aa < A ? 1.f : 0.f;
        }
        y[it] = th[idx] * m;
    }
    sched_fence();       // every load is issued before the first store
    mid();
    sched_fence();
    // scales (the kernels that feed a tanh are prescaled; the hidden_0 kernel also by w1s, the inverse of the power of two the
    // segment's observations are multiplied by) and the zero masks of the padding are applied here
    const float w1p = PROMP_TANH_PRESCALE * w1s;
    auto put = [&](const float (&x)[8], int b, float m0, float m1) {
        f32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = x[e] * m0;
            hi[e] = x[4 + e] * m1;
        }
        u32x4 pl[PROMP_NT];
        pass_split8(lo, hi, pl);
#pragma unroll
        for (int t = 0; t < PROMP_NT; ++t) sts_w4(sm + L.wp + t * L.plane_stride + 4 * (64 * b + lane), pl[t]);
    };
    // (a wave past the end of a kind stores the kind's last block a second time: same values, same addresses, no branch -- a
    //  guarded store invites the compiler to sink the block's loads into the guard, one more round trip)
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int bj = w + it * NW, b = bj < NK1 ? bj : NK1 - 1;
        f32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = x1[it][e] * (8 * kk + e < O ? w1p : 0.f);
            hi[e] = x1[it][4 + e] * (8 * kk + 4 + e < O ? w1p : 0.f);
        }
        u32x4 pl[PROMP_NT];
        pass_split8(lo, hi, pl);
#pragma unroll
        for (int t = 0; t < PROMP_NT; ++t) sts_w4(sm + L.wp + t * L.plane_stride + 4 * (64 * b + lane), pl[t]);
    }
#pragma unroll
    for (int it = 0; it < IT2; ++it) {
        const int bj = w + it * NW;
        put(x2[it], B1 + (bj < NK2 ? bj : NK2 - 1), PROMP_TANH_PRESCALE, PROMP_TANH_PRESCALE);
    }
#pragma unroll
    for (int it = 0; it < IT3; ++it) {
        const int bj = w + it * NW;
        put(x3[it], B2 + (bj < NK3 ? bj : NK3 - 1), 1.f, 1.f);
    }
#pragma unroll
    for (int it = 0; it < IT4; ++it) {
        const int bj = w + it * NW;
        put(x4[it], B3 + (bj < NK4 ? bj : NK4 - 1), aok ? 1.f : 0.f, aok ? 1.f : 0.f);
    }
#pragma unroll
    for (int it = 0; it < IS; ++it) {
        const int ej = tid + it * NT, e = ej < L.n_side ? ej : L.n_side - 1;
        sm[L.w3b + e] = y[it];
    }
}

// Cross-wave, fixed-order sum of the waves' gradient tiles -> one partial row in global memory (accumulators in the
// 32x32 / 16x16 result layouts of this kernel).  gsc / gsc1: the exact powers of two that undo this wave's cotangent scale (and, for
// the hidden_0 kernel, the observations' scale) on the way into the slab.  Every wave stores its tiles to its own LDS slab of [NP + 2] floats (every
// entry written exactly once), then all threads add the slabs in wave order.
template <int NC1, int NC2, int NW>
PROMP_DEV bool pass_reduce_to_partial(float* S, float* P, const f32x16 (&aw2)[NC1 / 2][NC2 / 2], const f32x16 (&aw1)[NC1 / 2],
                                      const f32x4 (&aw3)[NC2], const f32x4 (&gb1)[NC1], const f32x4 (&gb2)[NC2], float gs0,
                                      float gs1, float gb30, float gb31, float loss, float klsum, int O, int A, int tid, float gsc,
                                      float gsc1, int bad) {
    constexpr int H1 = 16 * NC1, H2 = 16 * NC2, NT = 64 * NW;
    tid += opaque_zero();         // (as in chain_reduce_to_partial: no lane-constant index kept alive, and spilled, across the tile loop)
    const int lane = tid & 63, w = tid >> 6, i16 = lane & 15, kk = lane >> 4, j32 = lane & 31, kh = lane >> 5;
    const int ob1 = O * H1, oW2 = ob1 + H1, ob2 = oW2 + H1 * H2, oW3 = ob2 + H2, ob3 = oW3 + H2 * A, oS = ob3 + A, NP = oS + A;
    const int SL = (NP + 2 + 3) & ~3;
    if (lane == 0) ((int*)S)[w - 4] = bad;     // the spare words in front of the slabs: this wave's overflow vote
    lds_barrier();                // every wave is done with the parameter planes / transposed tiles
    {
        float* mine = S + w * SL;
#pragma unroll
        for (int bi = 0; bi < NC1 / 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < NC2 / 2; ++bj)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    mine[oW2 + (32 * bi + (r & 3) + 8 * (r >> 2) + 4 * kh) * H2 + 32 * bj + j32] = aw2[bi][bj][r] * gsc;
#pragma unroll
        for (int bj = 0; bj < NC1 / 2; ++bj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;      // observation index
                if (row < O) mine[row * H1 + 32 * bj + j32] = aw1[bj][r] * gsc1;
            }
#pragma unroll
        for (int c = 0; c < NC2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (i16 < A) mine[oW3 + (16 * c + 4 * kk + r) * A + i16] = aw3[c][r] * gsc;
        if (i16 == 0) {           // (bias sums already folded over the 16 sample lanes) units 16 c + 4 kk + r; actions 2 kk, 2 kk + 1
#pragma unroll
            for (int c = 0; c < NC1; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[ob1 + 16 * c + 4 * kk + r] = gb1[c][r] * gsc;
#pragma unroll
            for (int c = 0; c < NC2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[ob2 + 16 * c + 4 * kk + r] = gb2[c][r] * gsc;
            if (2 * kk < A) {
                mine[ob3 + 2 * kk] = gb30;
                mine[oS + 2 * kk] = gs0;
            }
            if (2 * kk + 1 < A) {
                mine[ob3 + 2 * kk + 1] = gb31;
                mine[oS + 2 * kk + 1] = gs1;
            }
        }
        if (lane == 0) {
            mine[NP] = loss;
            mine[NP + 1] = klsum;
        }
    }
    lds_barrier();
    // four entries per thread and trip (16-byte LDS reads and one 16-byte store: a quarter of the instructions of the
    // entry-by-entry loop, 4.6 k -> measured cycles per segment); the slabs and the partial row are padded to SL floats, the pad is
    // summed and stored like the rest and never read
#pragma unroll 2
    for (int e = 4 * tid; e < SL; e += 4 * NT) {
        f32x4 v[NW];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) v[ww] = *(const f32x4*)(S + ww * SL + e);      // all slab reads in flight together
        f32x4 t = v[0];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) t += v[ww];
        *(f32x4*)(P + e) = t;
    }
    int any = 0;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) any |= ((const int*)S)[ww - 4];
    return any != 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The tile walk.  One wave per SIMD issues in order and a vector instruction occupies it for four cycles, so the wave is bound
// by what it issues (1.3 k instructions per tile) plus whatever it waits for.  Measured on the first version of this kernel:
// a third of all wave cycles parked on s_waitcnt -- every GEMM phase started by loading its operand fragments and waiting.
// (Running the backward half of one tile against the forward half of the next -- two instruction streams in one basic block,
// forced together with sched_group_barrier -- was measured too: no faster, the lone wave's issue rate is the limit either way,
// and the carried planes cost 260 accumulator-register moves per tile.)  The tile is therefore walked front to back in eleven
// regions separated by scheduling barriers, and every region REQUESTS the operands of the next one: weight fragments one GEMM
// ahead, the transpose reads of a weight gradient as soon as its tiles are written -- under the GEMM in between.  A lane
// has 512 registers at one wave per SIMD; about 130 of them hold operands in flight.  Nothing in the loop is a memory
// barrier: the order of a tile's LDS writes and transpose reads is a memory dependence the compiler keeps (same base,
// run-time offsets), and the epilogue is branch free.
// ---------------------------------------------------------------------------------------------------------------------------
template <int NC1, int NC2>
struct PassSums {                // what a wave accumulates over its tiles of a segment
    f32x16 aw2[NC1 / 2][NC2 / 2], aw1[NC1 / 2];
    f32x4 aw3[NC2], gb1[NC1], gb2[NC2];
    float loss, klsum, gs0, gs1, gb30, gb31;
    // FP16 split: the power of two 2^ck this wave's cotangents are multiplied by in the segment (cs), its inverse (ics); prov: no tile
    // has had a nonzero cotangent yet (the accumulators are exactly zero and the scale is still free)
    float cs, ics, amax;         // amax: the largest |mean cotangent| the wave has met in the segment, unscaled
    int prov;
};
// FP16 has a range.  From the distribution epilogue on, a tile's cotangents -- and everything the backward half accumulates --
// carry the wave's power of two S.cs, undone on the way into the end-of-segment slab.  The wave's first tile with a cotangent sets
// it (until then the accumulators are exactly zero and the scale is free): the largest mean cotangent of that tile goes to
// [2^t, 2^(t+1)), t = PASS_CT_TARGET: the typical cotangent of every backward stage then stays above 2^-3 (where a value's low term
// turns subnormal and its error stops shrinking with the value: 2^-25 absolute) for layer gains down to ~2^-6, and later tiles have
// 2^6 of headroom (times 2^5 of backward gain) under FP16's 65504.  The tile itself pays one instruction for this: the wave's running
// maximum S.amax.  Cotangents are heavy-tailed when the policy has moved far from the one that sampled (the ratio is an exponential);
// a wave whose maximum has left the format at its scale (the mean cotangents' planes feed the output kernel's sums only), or that finds
// an infinity or a NaN in its hidden_0 sums (every other split ends there), says so at the end of the segment, and the workgroup
// walks the segment again -- now every wave knows the largest cotangent it will meet and puts THAT at 2^PASS_CT_REDO (a third walk,
// for backward gains beyond 2^5: PASS_CT_RETRY lower).  One such segment doubles the launch's duration: the other workgroups wait.  Guarded in-tile variants (rescaling
// the sums in place; abandoning the tile and walking it again) were built and measured: +3.4 / +4.7 us per launch for an event that
// does not occur at PPO's operating point.
#ifndef PROMP_CT_ATTEMPTS
#define PROMP_CT_ATTEMPTS 3
#endif
#ifndef PROMP_PASS_CT_TARGET
#define PROMP_PASS_CT_TARGET 5
#endif
PROMP_CX int PASS_CT_TARGET = PROMP_PASS_CT_TARGET, PASS_CT_REDO = 10, PASS_CT_RETRY = 12, PASS_CT_ATTEMPTS = PROMP_CT_ATTEMPTS;
// the scale a largest |mean cotangent| of mx asks for (mx = 0 / not finite: 2^-4 N, for adv / N, and `prov`)
template <int NC1, int NC2>
PROMP_DEV void pass_cotangent_scale(PassSums<NC1, NC2>& S, float mx, float invN, int target) {
    const bool okm = mx > 0.f && mx < 3.0e38f;
    int k = scale_exp(okm ? mx : invN, okm ? target : -4);
    k = k < -100 ? -100 : k > 100 ? 100 : k;
    S.cs = pow2f(k);
    S.ics = pow2f(-k);
    S.prov = okm ? 0 : 1;
}

// constants of a wave's walk through a segment
struct PassWalk {
    const float *obs, *act, *adv, *old_mean, *old_log_std;
    float* hcache;
    int ls_per_row, O, A, task, trow0, tnrows, tend, loss_kind;
    int ct_target;               // FP16 split: where the first tile's largest cotangent goes (pass_cotangent_scale)
    float invN, clip_eps, sums, xs;      // xs: the power of two the task's observations are multiplied by (FP16 split)
    float s0, s1, e0, e1, sn20, sn21, rden0, rden1;
    int q0, q1;
    bool own0, own1;
    unsigned long long* dbg;     // developer tooling: cycle stamps (NULL unless a -DPROMP_DEV_STAMPS build asks for them)
    int tix;
};
#ifdef PROMP_DEV_STAMPS
#define PASS_STAMP(j) do { if (W.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) W.dbg[8 + 16 * (W.tix < 3 ? W.tix : 3) + (j)] = promp_clock(); } while (0)
#else
#define PASS_STAMP(j) do { } while (0)
#endif
// this lane's eight observation entries of its sample in tile t: obs[row i16][8 kk .. 8 kk + 7] (zeros outside the tile / task)
PROMP_DEV void pass_load_x(float (&xr)[8], const PassWalk& W, int t, int i16, int kk) {
    const int nv = (t < W.tend) ? (W.tnrows - 16 * t < 16 ? W.tnrows - 16 * t : 16) : 0;
    const bool rv = i16 < nv;
    const float* src = W.obs + ((long long)W.trow0 + (t < W.tend ? 16 * t : 0) + (rv ? i16 : 0)) * W.O;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int o = 8 * kk + e;
        const bool ok = rv && o < W.O;
        xr[e] = src[ok ? o : 0] * (ok ? W.xs : 0.f);
    }
}

// acc[c] += sum over the products (ta + tb < NT) of (A-side fragments wf[ta][c]) x (B-side planes xb[tb])
template <int NC>
PROMP_DEV void pass_gemm16(f32x4 (&acc)[NC], const u32x4 (&wf)[PROMP_NT][NC], const u32x4 (&xb)[PROMP_NT]) {
#pragma unroll
    for (int ta = PROMP_NT - 1; ta >= 0; --ta)
#pragma unroll
        for (int tb = PROMP_NT - 1 - ta; tb >= 0; --tb)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = mfma16_sw<PROMP_NT>(wf[ta][c], xb[tb], acc[c]);
}
// the fragments [term][c] of NC consecutive 64-fragment blocks starting at `first`, stride `cs` blocks between them
template <int NC>
PROMP_DEV void pass_load_frags(u32x4 (&wf)[PROMP_NT][NC], const u32x4* F, int PS, int first, int cs) {
#pragma unroll
    for (int ta = 0; ta < PROMP_NT; ++ta)
#pragma unroll
        for (int c = 0; c < NC; ++c) wf[ta][c] = F[ta * PS + first + c * cs];
}
// One 16-sample tile, front to back.  `w1f` arrives loaded (layer 1's fragments, requested at the end of the previous tile)
// and leaves requested for the next tile; `xr` likewise (the observations).
// One tile's pending hidden_0 kernel gradient: the operand planes (already in registers) of aw1 += X^T dZ1.  Its matrix
// instructions are issued in the NEXT tile's first tanh / split region, whose vector work runs in their shadow.
template <int NC1>
struct PassPending {
    u32x4 fx[PROMP_NT], fd[NC1 / 2][PROMP_NT];
};
template <int NC1, int NC2>
PROMP_DEV void pass_flush_pending(PassSums<NC1, NC2>& S, const PassPending<NC1>& Q) {
#pragma unroll
    for (int ta = PROMP_NT - 1; ta >= 0; --ta)
#pragma unroll
        for (int tb = PROMP_NT - 1 - ta; tb >= 0; --tb)
#pragma unroll
            for (int bj = 0; bj < NC1 / 2; ++bj) S.aw1[bj] = mfma32_sw<PROMP_NT>(Q.fx[ta], Q.fd[bj][tb], S.aw1[bj]);
}

template <int NC1, int NC2, bool BWD, bool STORE, bool PENDING>
PROMP_DEV void pass_tile(PassSums<NC1, NC2>& S, PassPending<NC1>& Q, u32x4 (&w1f)[PROMP_NT][NC1], float (&xr)[8], const PassWalk& W,
                         const PassTileAddr& T, float* sm, float* wreg, int lane, int t, int tnext) {
    constexpr int H1 = 16 * NC1, H2 = 16 * NC2, NP1 = NC1 / 2, NP2 = NC2 / 2, NB1 = NC1 / 2, NB2 = NC2 / 2;
    constexpr int HCR = chain_cache_row(H1, H2);
    constexpr int TPL = PROMP_PASS_TPLANE, XPL = PROMP_PASS_XPLANE, DPL = PROMP_PASS_DPLANE;
    constexpr PassLds L = pass_layout(NC1, NC2, 1, 0);
    constexpr int PS = L.n_frag;
    const int i16 = lane & 15, kk = lane >> 4;
    const u32x4* F = (const u32x4*)(sm + L.wp) + lane;
    float *XT = wreg + L.xt, *TA = wreg + L.ta, *TB = wreg + L.tb, *DM = wreg + L.dm;
    const int nrows = (W.tnrows - 16 * t) < 16 ? (W.tnrows - 16 * t) : 16;
    const long long base = (long long)W.trow0 + 16 * t;
    const bool rvalid = i16 < nrows;
    const long long n = base + (rvalid ? i16 : 0);
    float* const hcb = STORE ? W.hcache + (base + 16 * W.task) * HCR + 16 * i16 + 4 * kk : nullptr;

    // ---- region 0: row data, observation planes, layer 1; requests layer 2's first fragments
    PASS_STAMP(0);
    const float* olsp = W.old_log_std + (W.ls_per_row ? n * W.A : (long long)W.task * W.A);
    const float advn = W.adv[n] * (rvalid ? 1.f : 0.f);
    const float ac0 = W.act[n * W.A + W.q0], ac1 = W.act[n * W.A + W.q1];
    const float mo0 = W.old_mean[n * W.A + W.q0], mo1 = W.old_mean[n * W.A + W.q1];
    const float so0 = olsp[W.q0], so1 = olsp[W.q1];
    u32x4 xB[PROMP_NT];
    {
        f32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = xr[e];
            hi[e] = xr[4 + e];
        }
        pass_split8(lo, hi, xB);
    }
    pass_load_x(xr, W, tnext, i16, kk);
    if (BWD) {
#pragma unroll
        for (int tt = 0; tt < PROMP_NT; ++tt) {
            sts_w2(XT + tt * XPL + T.xw0, xB[tt][0], xB[tt][1]);
            sts_w2(XT + tt * XPL + T.xw1, xB[tt][2], xB[tt][3]);
        }
    }
    f32x4 h1[NC1], h2[NC2];
    {
        const float* B1l = sm + L.b1 + 4 * kk;
#pragma unroll
        for (int c = 0; c < NC1; ++c) h1[c] = lds4(B1l + 16 * c);
    }
    pass_gemm16<NC1>(h1, w1f, xB);                               // Z1^T = W1^T X^T + b1 (K = 32 observation slots, zero padded)
    u32x4 w2f[NP1][PROMP_NT][NC2];
    pass_load_frags<NC2>(w2f[0], F, PS, L.f_w2f, NP1 * 64);      // [c2][P = 0]
    // (the split first; then layer 1's matrix instructions with the region's address arithmetic and requests in their shadow)
    PROMP_SCHED_VALU(PROMP_NT == 3 ? 48 : 20);
#pragma unroll
    for (int i = 0; i < PROMP_NPROD * NC1; ++i) {
        PROMP_SCHED_MFMA(1);
        PROMP_SCHED_VALU(3);
        PROMP_SCHED_DSREAD(1);
    }
    sched_fence();
    // ---- region 1: tanh, hidden_0 planes (-> first tile); requests the rest of layer 2
    PASS_STAMP(1);
    u32x4 hB1[NP1][PROMP_NT];
#pragma unroll
    for (int c = 0; c < NC1; ++c) {
        h1[c] = pass_tanh4(h1[c]);
        if (STORE) *(f32x4*)(hcb + 256 * c) = h1[c];
    }
#pragma unroll
    for (int P = 0; P < NP1; ++P) {
        pass_split8(h1[2 * P], h1[2 * P + 1], hB1[P]);
        if (BWD) pass_store_planes(TA, TPL, T.wr + 256 * P, hB1[P]);
    }
#pragma unroll
    for (int P = 1; P < NP1; ++P) pass_load_frags<NC2>(w2f[P], F, PS, L.f_w2f + P * 64, NP1 * 64);
    if (BWD && PENDING) {
        // the previous tile's hidden_0 kernel gradient: 6 NB1 matrix instructions of 32 cycles, this region's vector work in between
        pass_flush_pending<NC1, NC2>(S, Q);
#pragma unroll
        for (int i = 0; i < PROMP_NPROD * NB1; ++i) {
            PROMP_SCHED_MFMA(1);
            PROMP_SCHED_VALU(PROMP_NT == 3 ? 7 : 14);
        }
    }
    sched_fence();
    // ---- region 2: layer 2; requests the output layer's fragments
    PASS_STAMP(2);
    {
        const float* B2l = sm + L.b2 + 4 * kk;
#pragma unroll
        for (int c = 0; c < NC2; ++c) h2[c] = lds4(B2l + 16 * c);
    }
#pragma unroll
    for (int P = 0; P < NP1; ++P) pass_gemm16<NC2>(h2, w2f[P], hB1[P]);
    u32x4 w3f[NP2][PROMP_NT][1];
#pragma unroll
    for (int P = 0; P < NP2; ++P) pass_load_frags<1>(w3f[P], F, PS, L.f_w3f + P * 64, 0);
    sched_fence();
    // ---- region 3: tanh, hidden_1 planes (-> second tile)
    PASS_STAMP(3);
    u32x4 hB2[NP2][PROMP_NT];
#pragma unroll
    for (int c = 0; c < NC2; ++c) {
        h2[c] = pass_tanh4(h2[c]);
        if (STORE) *(f32x4*)(hcb + 256 * (NC1 + c)) = h2[c];
    }
#pragma unroll
    for (int P = 0; P < NP2; ++P) {
        pass_split8(h2[2 * P], h2[2 * P + 1], hB2[P]);
        if (BWD) pass_store_planes(TB, TPL, T.wr + 256 * P, hB2[P]);
    }
    sched_fence();
    // ---- region 4: output layer mu^T = W3^T H2^T + b3 (rows of the product: action slots, see pass_stage_net; two accumulators
    //      in turn), distribution + objective (branch free), mean cotangents (-> their tile); requests the output-kernel
    //      gradient's operands
    PASS_STAMP(4);
    float mu0, mu1;
    {
        const float* B3l = sm + L.b3 + 2 * kk;
        f32x4 m[2] = {zero4(), zero4()};
        const f32x2 bb = lds2(B3l);
        m[0][0] = bb[0];
        m[0][1] = bb[1];
#pragma unroll
        for (int P = 0; P < NP2; ++P)
#pragma unroll
            for (int ta = PROMP_NT - 1; ta >= 0; --ta)
#pragma unroll
                for (int tb = PROMP_NT - 1 - ta; tb >= 0; --tb)
                    m[(ta + tb + P) & 1] = mfma16_sw<PROMP_NT>(w3f[P][ta][0], hB2[P][tb], m[(ta + tb + P) & 1]);
        mu0 = m[0][0] + m[1][0];
        mu1 = m[0][1] + m[1][1];
        if (STORE) {
            f32x2 mm;
            mm[0] = mu0;
            mm[1] = mu1;
            *(f32x2*)(W.hcache + (base + 16 * W.task) * HCR + 256 * (NC1 + NC2) + 8 * i16 + 2 * kk) = mm;
        }
    }
    float d0, d1;
    {   // lane (i16, kk) = sample i16, actions 2 kk and 2 kk + 1
        const float o0 = W.own0 ? 1.f : 0.f, o1 = W.own1 ? 1.f : 0.f, rv = rvalid ? 1.f : 0.f;
        const float z0 = (ac0 - mu0) * W.e0, z1 = (ac1 - mu1) * W.e1;
        const float zo0 = (ac0 - mo0) * fast_exp(-so0), zo1 = (ac1 - mo1) * fast_exp(-so1);
        const float num0 = (mo0 - mu0) * (mo0 - mu0) + fast_exp(2.f * so0) - W.sn20;
        const float num1 = (mo1 - mu1) * (mo1 - mu1) + fast_exp(2.f * so1) - W.sn21;
        const float den0 = 2.f * W.sn20 + 1e-8f, den1 = 2.f * W.sn21 + 1e-8f;
        float dlp = o0 * ((so0 - W.s0) - 0.5f * (z0 * z0 - zo0 * zo0)) + o1 * ((so1 - W.s1) - 0.5f * (z1 * z1 - zo1 * zo1));
        float sumz2 = o0 * (z0 * z0) + o1 * (z1 * z1);
        float kl = o0 * (num0 * W.rden0 + W.s0 - so0) + o1 * (num1 * W.rden1 + W.s1 - so1);
        dlp = fold_groups16(dlp);          // sums over the row's actions (the four lane groups)
        sumz2 = fold_groups16(sumz2);
        kl = fold_groups16(kl);
        const float rho = expf(rvalid ? dlp : 0.f);        // (padding rows: a finite ratio with zero weight)
        const float aw = advn * W.invN;
        const float x = rho * advn, y = fminf(fmaxf(rho, 1.f - W.clip_eps), 1.f + W.clip_eps) * advn;
        const float lp = -W.sums - 0.5f * sumz2 - 0.5f * (float)W.A * 1.8378770664093453f;
        const bool is_kl = W.loss_kind == LOSS_KL, is_ratio = W.loss_kind == LOSS_RATIO, is_clip = W.loss_kind == LOSS_CLIP;
        // d loss / d logpi (c), the weight of the KL cotangents (ck; LOSS_KL only), the row's objective term
        const float c = is_kl ? 0.f : is_ratio ? -aw * rho : is_clip ? ((x <= y) ? -aw * rho : 0.f) : -aw;
        const float ck = is_kl ? rv * W.invN : 0.f;
        const float lrow = is_kl ? kl * W.invN : is_ratio ? -rho * aw : is_clip ? -fminf(x, y) * W.invN : -lp * aw;
        const float first = (kk == 0) ? rv : 0.f;          // one lane per row carries the row's scalars
        const float dklm0 = -2.f * (mo0 - mu0) * W.rden0, dklm1 = -2.f * (mo1 - mu1) * W.rden1;
        const float dkls0 = (-2.f * W.sn20 * den0 - 4.f * num0 * W.sn20) * (W.rden0 * W.rden0) + 1.f;
        const float dkls1 = (-2.f * W.sn21 * den1 - 4.f * num1 * W.sn21) * (W.rden1 * W.rden1) + 1.f;
        d0 = o0 * (c * z0 * W.e0 + ck * dklm0);
        d1 = o1 * (c * z1 * W.e1 + ck * dklm1);
        if (BWD && PROMP_NT == 2) {
            const float am = fmaxf(fabsf(d0), fabsf(d1));
            S.amax = fmaxf(S.amax, am);
            if (wave_uniform(S.prov)) pass_cotangent_scale<NC1, NC2>(S, wave_absmax_f32(am), W.invN, W.ct_target);
        }
        S.loss += first * lrow;
        S.klsum += first * (kl * W.invN);
        S.gs0 += o0 * (c * (z0 * z0 - 1.f) + ck * dkls0);
        S.gs1 += o1 * (c * (z1 * z1 - 1.f) + ck * dkls1);
        S.gb30 += d0;
        S.gb31 += d1;
        if (BWD && PROMP_NT == 2) {
            d0 *= S.cs;
            d1 *= S.cs;
        }
    }
    if (!BWD) {
        pass_load_frags<NC1>(w1f, F, PS, L.f_w1, 64);            // the next tile's layer 1
        sched_fence();
        return;
    }
    {
        unsigned dw[PROMP_NT];
        split_pair<PROMP_NT>(d0, d1, dw);
#pragma unroll
        for (int tt = 0; tt < PROMP_NT; ++tt) DM[tt * DPL + T.dmw] = __builtin_bit_cast(float, dw[tt]);
    }
    u32x4 bD[PROMP_NT], aH[NC2][PROMP_NT];
    pass_read_tr(bD, DM, DPL, T.dr0, T.dr1);
#pragma unroll
    for (int c = 0; c < NC2; ++c) pass_read_tr(aH[c], TB, TPL, T.rd16_0 + 2 * (128 * (c >> 1) + 16 * (c & 1)), T.rd16_1 + 2 * (128 * (c >> 1) + 16 * (c & 1)));
    f32x2 wb[NC2];
    {
        const float* W3b = sm + L.w3b + lane * 2;
#pragma unroll
        for (int c = 0; c < NC2; ++c) wb[c] = lds2(W3b + c * 128);
    }
    sched_fence();
    // ---- region 5: dH2^T = W3 dmu^T (K = act_dim:
    //      the exact FP32 instruction, k-step ro <-> actions 2 kk + ro); requests the first backward hidden_1 fragments
    PASS_STAMP(5);
    f32x4 dz2[NC2];
#pragma unroll
    for (int c = 0; c < NC2; ++c) dz2[c] = mfma16(wb[c][0], d0, zero4());
#pragma unroll
    for (int c = 0; c < NC2; ++c) dz2[c] = mfma16(wb[c][1], d1, dz2[c]);
    u32x4 w2b[NP2][PROMP_NT][NC1];
    pass_load_frags<NC1>(w2b[0], F, PS, L.f_w2b, NP2 * 64);      // [c1][P = 0]
    sched_fence();
    // ---- region 6: dZ2^T = dH2^T * (1 - H2^2), its planes (-> second tile, over the hidden_1 planes)
    PASS_STAMP(6);
    u32x4 dB2[NP2][PROMP_NT];
#pragma unroll
    for (int c = 0; c < NC2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dz2[c][r] *= -pass_neg_dtanh(h2[c][r]);
            S.gb2[c][r] += dz2[c][r];
        }
#pragma unroll
    for (int P = 0; P < NP2; ++P) {
        pass_split8(dz2[2 * P], dz2[2 * P + 1], dB2[P]);
        pass_store_planes(TB, TPL, T.wr + 256 * P, dB2[P]);      // (after the output-kernel gradient's reads: a memory dependence)
    }
    // the output-kernel gradient aw3[unit][action] += sum_s H2[s][unit] dmu[s][action] (operands read in region 4) in this
    // region's shadow: 6 NC2 matrix instructions of 16 cycles, three vector instructions after each
#pragma unroll
    for (int ta = PROMP_NT - 1; ta >= 0; --ta)
#pragma unroll
        for (int tb = PROMP_NT - 1 - ta; tb >= 0; --tb)
#pragma unroll
            for (int c = 0; c < NC2; ++c) S.aw3[c] = mfma16_sw<PROMP_NT>(aH[c][ta], bD[tb], S.aw3[c]);
#pragma unroll
    for (int i = 0; i < PROMP_NPROD * NC2; ++i) {
        PROMP_SCHED_MFMA(1);
        PROMP_SCHED_VALU(PROMP_NT == 3 ? 4 : 6);
        PROMP_SCHED_DSWRITE(1);
    }
    sched_fence();
    // ---- region 7: dH1^T = W2 dZ2^T; requests the rest of its fragments (under its first half) and the planes of the hidden_1
    //      and hidden_0 kernel gradients (hidden_0 activations, dZ2, observations: their tiles are complete)
    PASS_STAMP(7);
#pragma unroll
    for (int P = 1; P < NP2; ++P) pass_load_frags<NC1>(w2b[P], F, PS, L.f_w2b + P * 64, NP2 * 64);
    u32x4 fa[NB1][PROMP_NT], fb[NB2][PROMP_NT];
#pragma unroll
    for (int b = 0; b < NB1; ++b) pass_read_tr(fa[b], TA, TPL, T.rd32_0 + 256 * b, T.rd32_1 + 256 * b);
#pragma unroll
    for (int b = 0; b < NB2; ++b) pass_read_tr(fb[b], TB, TPL, T.rd32_0 + 256 * b, T.rd32_1 + 256 * b);
    f32x4 ad1[NC1];
#pragma unroll
    for (int c = 0; c < NC1; ++c) ad1[c] = zero4();
#pragma unroll
    for (int P = 0; P < NP2; ++P) pass_gemm16<NC1>(ad1, w2b[P], dB2[P]);
    if (STORE) {
#pragma unroll
        for (int c = 0; c < NC1; ++c) *(f32x4*)(hcb + 256 * (NC1 + NC2) + 128 + 256 * c) = PROMP_NT == 2 ? ad1[c] * S.ics : ad1[c];
    }
    u32x4 fx[PROMP_NT];
    pass_read_tr(fx, XT, XPL, T.rd32_0, T.rd32_1);
    sched_fence();
    // ---- region 8: dZ1^T = dH1^T * (1 - H1^2), its planes (-> first tile, over the hidden_0 planes: their reads were issued in
    //      region 7)
    PASS_STAMP(8);
#pragma unroll
    for (int c = 0; c < NC1; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ad1[c][r] *= -pass_neg_dtanh(h1[c][r]);
            S.gb1[c][r] += ad1[c][r];
        }
#pragma unroll
    for (int P = 0; P < NP1; ++P) {
        u32x4 dB1[PROMP_NT];
        pass_split8(ad1[2 * P], ad1[2 * P + 1], dB1);
        pass_store_planes(TA, TPL, T.wr + 256 * P, dB1);
    }
    // the hidden_1 kernel gradient aw2[u1][u2] += sum_s H1[s][u1] dZ2[s][u2] on 32x32x16 (operands read in regions 5 / 6) in this
    // region's shadow: 6 NB1 NB2 matrix instructions of 32 cycles, seven vector / LDS instructions after each
#pragma unroll
    for (int ta = PROMP_NT - 1; ta >= 0; --ta)
#pragma unroll
        for (int tb = PROMP_NT - 1 - ta; tb >= 0; --tb)
#pragma unroll
            for (int bi = 0; bi < NB1; ++bi)
#pragma unroll
                for (int bj = 0; bj < NB2; ++bj) S.aw2[bi][bj] = mfma32_sw<PROMP_NT>(fa[bi][ta], fb[bj][tb], S.aw2[bi][bj]);
#pragma unroll
    for (int i = 0; i < PROMP_NPROD * NB1 * NB2; ++i) {
        PROMP_SCHED_MFMA(1);
        PROMP_SCHED_VALU(PROMP_NT == 3 ? 6 : 7);
        PROMP_SCHED_DSWRITE(1);
    }
    sched_fence();
    // ---- region 9: the hidden_0 kernel gradient aw1[obs][u1] += sum_s X[s][obs] dZ1[s][u1] (one 32-block of observation slots)
    //      is left pending: its planes are requested here, its matrix instructions run under the next tile's region 1; requests
    //      the next tile's layer 1
    PASS_STAMP(9);
#pragma unroll
    for (int tt = 0; tt < PROMP_NT; ++tt) Q.fx[tt] = fx[tt];
#pragma unroll
    for (int b = 0; b < NB1; ++b) pass_read_tr(Q.fd[b], TA, TPL, T.rd32_0 + 256 * b, T.rd32_1 + 256 * b);
    pass_load_frags<NC1>(w1f, F, PS, L.f_w1, 64);
    sched_fence();
    PASS_STAMP(10);
}

// BWD = false: objective and mean KL only (compute_stats / line-search evaluations): the tile walk stops after the
// distribution epilogue and the partial row carries just the two scalars.
// STORE: the hidden activations, the means and the hidden_0 cotangent before its tanh derivative (W2 dZ2^T) of every tile also
// go to the step's primal cache (chain_cache_row in promp_kernels_chain.h) for the R-operator pass that follows at the same
// parameters; the register-chain layout IS that pass's operand order, so a block is 3 NC1 + NC2 16-byte stores + one 8-byte.
template <int NC1, int NC2, int NW, bool BWD, bool STORE>
__global__ void __launch_bounds__(64 * NW, NW / 4) k_pass(PassArgs a) {
    constexpr int H1 = 16 * NC1, H2 = 16 * NC2, NB1 = NC1 / 2, NB2 = NC2 / 2;
    constexpr int DPL = PROMP_PASS_DPLANE;
    constexpr PassLds L = pass_layout(NC1, NC2, NW, 0);
    PROMP_SMEM_DECL;
    float* sm = (float*)PROMP_SMEM_PTR;
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
    const int i16 = lane & 15, kk = lane >> 4;
    const int O = a.O, A = a.A;
    const int ob1 = O * H1, oW2 = ob1 + H1, ob2 = oW2 + H1 * H2, oW3 = ob2 + H2, ob3 = oW3 + H2 * A, oS = ob3 + A, NP = oS + A;
    const float* dist = sm + L.dist;
    float* wreg = sm + L.wave0 + w * L.wave_stride;
    const PassTileAddr T = pass_tile_addr(lane);
    PassWalk W;
    W.obs = a.obs; W.act = a.act; W.adv = a.adv; W.old_mean = a.old_mean; W.old_log_std = a.old_log_std; W.hcache = a.hcache;
    W.ls_per_row = a.ls_per_row; W.O = O; W.A = A; W.loss_kind = a.loss_kind; W.clip_eps = a.clip_eps;
    W.dbg = a.dbg; W.tix = 0;
    W.own0 = 2 * kk < A; W.own1 = 2 * kk + 1 < A;
    W.q0 = W.own0 ? 2 * kk : 0; W.q1 = W.own1 ? 2 * kk + 1 : 0;

    const int wgi = PROMP_PASS_XCD ? xcd_item(blockIdx.x, gridDim.x) : (int)blockIdx.x;     // (see PROMP_PASS_XCD)
    const int sg0 = a.wg_seg_offsets[wgi], sg1 = a.wg_seg_offsets[wgi + 1];
    CH_WGSTAMP(0);
    int attempt = 0;             // FP16 split: how often the current segment has overflowed (see the end of the tile walk)
    float redo_amax = 0.f;       // ... and the largest cotangent this wave met on the way
    for (int sg = sg0; sg < sg1; ++sg) {
        const ChainSeg seg = a.segs[sg];
        W.task = seg.task;
        W.trow0 = a.task_row_offsets[seg.task];
        W.tnrows = a.task_row_offsets[seg.task + 1] - W.trow0;
        W.invN = 1.0f / (float)W.tnrows;
        W.tend = seg.tile0 + seg.ntiles;
        const float* th = a.theta + (long long)seg.task * a.theta_task_stride;
        // the segment's requests in the order their answers are needed: the parameters (from L2) first, the first tile's
        // observations (from memory) behind the staging's last load; the workgroup joins after everything is on its way
        float xr[8];
        int t = seg.tile0 + w;
        const ChainDistRaw draw = chain_dist_load(th, nullptr, oS, A, tid);
        CH_STAMP(0);
        // FP16 split: the task's observations times 2^-sx, the hidden_0 kernel times 2^sx (obs_shift, promp_kernels_chain.h)
        float w1s = 1.f;
        W.xs = 1.f;
        {
            const int sx = obs_shift(a.obs_absmax, seg.task);
            W.xs = pow2f(-sx);
            w1s = pow2f(sx);
        }
        pass_stage_net<NC1, NC2, NW>(sm, th, O, A, tid, w1s, [&]() {
            pass_load_x(xr, W, t, i16, kk);
            __syncthreads();
        });
        chain_stage_dist(sm + L.dist, draw, A, a.clip_log_std, a.min_log_std, tid);
        // the action slots >= 8 of the cotangent tiles read as zero (the end-of-segment slabs alias them: once per segment)
        for (int e = lane; e < PROMP_NT * DPL; e += 64) wreg[L.dm + e] = 0.f;
        __syncthreads();
        CH_STAMP(1);
        W.s0 = dist[CH_LS + W.q0]; W.s1 = dist[CH_LS + W.q1]; W.e0 = dist[CH_ES + W.q0]; W.e1 = dist[CH_ES + W.q1];
        W.sn20 = dist[CH_SN2 + W.q0]; W.sn21 = dist[CH_SN2 + W.q1]; W.rden0 = dist[CH_RDEN + W.q0]; W.rden1 = dist[CH_RDEN + W.q1];
        W.sums = 0.f;                                 // sum of the log standard deviations (log-likelihood objective)
        for (int aa = 0; aa < A; ++aa) W.sums += dist[CH_LS + aa];

        PassSums<NC1, NC2> S;
#pragma unroll
        for (int i = 0; i < NB1; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) S.aw1[i][r] = 0.f;
#pragma unroll
            for (int j = 0; j < NB2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) S.aw2[i][j][r] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < NC2; ++c) {
            S.aw3[c] = zero4();
            S.gb2[c] = zero4();
        }
#pragma unroll
        for (int c = 0; c < NC1; ++c) S.gb1[c] = zero4();
        S.loss = S.klsum = S.gs0 = S.gs1 = S.gb30 = S.gb31 = 0.f;
        S.cs = S.ics = 1.f;
        S.amax = 0.f;
        S.prov = 1;
        W.ct_target = PASS_CT_TARGET;
        if (PROMP_NT == 2 && attempt > 0) {      // the segment again: the wave's largest cotangent is known
            pass_cotangent_scale<NC1, NC2>(S, redo_amax, W.invN, PASS_CT_REDO - (attempt - 1) * PASS_CT_RETRY);
            W.ct_target = PASS_CT_REDO - (attempt - 1) * PASS_CT_RETRY;      // (a wave that met no cotangent: still provisional)
        }

        u32x4 w1f[PROMP_NT][NC1];
        pass_load_frags<NC1>(w1f, (const u32x4*)(sm + L.wp) + lane, L.n_frag, L.f_w1, 64);
        W.tix = 0;
        PassPending<NC1> Q;
        if (t < W.tend) {
            pass_tile<NC1, NC2, BWD, STORE, false>(S, Q, w1f, xr, W, T, sm, wreg, lane, t, t + NW);
            for (t += NW; t < W.tend; t += NW) {
                W.tix += 1;
                pass_tile<NC1, NC2, BWD, STORE, true>(S, Q, w1f, xr, W, T, sm, wreg, lane, t, t + NW);
            }
            if (BWD) pass_flush_pending<NC1, NC2>(S, Q);
        }
        // FP16 split (pass_cotangent_scale): did this wave's cotangents stay inside the format?  Its largest one at its scale, and --
        // for backward gains beyond the bound -- an infinity or a NaN in the hidden_0 bias / kernel sums, the last links of the
        // chain (x * 0 is 0 for finite x only).  The waves vote inside pass_reduce_to_partial (its first barrier); a workgroup with
        // an overflow walks the segment again, every wave with the largest cotangent it met (redo_amax) for a scale.
        int bad = 0;
        if (BWD && PROMP_NT == 2 && attempt + 1 < PASS_CT_ATTEMPTS) {
            float chk = 0.f;
#pragma unroll
            for (int c = 0; c < NC1; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) chk = __builtin_fmaf(S.gb1[c][r], 0.f, chk);
#pragma unroll
            for (int i = 0; i < NB1; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) chk = __builtin_fmaf(S.aw1[i][r], 0.f, chk);
            redo_amax = wave_absmax_f32(S.amax);
            bad = (wave_any(chk != chk) || !(redo_amax * S.cs <= 65504.f)) ? 1 : 0;
        }
        CH_STAMP(2);

        float* P = a.partials + (long long)sg * a.partial_stride;
        float loss = S.loss, klsum = S.klsum, gs0 = S.gs0, gs1 = S.gs1, gb30 = S.gb30, gb31 = S.gb31;
        // per-action sums: lanes of one kk group differ in the sample; scalars live in the kk = 0 lanes
        gs0 = row16_sum(gs0);  gs1 = row16_sum(gs1);  gb30 = row16_sum(gb30);
        gb31 = row16_sum(gb31);  loss = row16_sum(loss);  klsum = row16_sum(klsum);
        if (!BWD) {   // only the two scalars leave the workgroup
            lds_barrier();
            if (lane == 0) {
                sm[4 + 2 * w] = loss;
                sm[4 + 2 * w + 1] = klsum;
            }
            lds_barrier();
            if (tid == 0) {
                float l = 0.f, k = 0.f;
                for (int ww = 0; ww < NW; ++ww) {
                    l += sm[4 + 2 * ww];
                    k += sm[4 + 2 * ww + 1];
                }
                P[NP] = l;
                P[NP + 1] = k;
            }
            continue;
        }
#pragma unroll
        for (int c = 0; c < NC1; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) S.gb1[c][r] = row16_sum(S.gb1[c][r]);
#pragma unroll
        for (int c = 0; c < NC2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) S.gb2[c][r] = row16_sum(S.gb2[c][r]);
        gs0 *= dist[CH_LMASK + W.q0];
        gs1 *= dist[CH_LMASK + W.q1];
        const bool redo = pass_reduce_to_partial<NC1, NC2, NW>(sm + 4, P, S.aw2, S.aw1, S.aw3, S.gb1, S.gb2, gs0, gs1, gb30, gb31, loss, klsum, O, A, tid,
                                                               S.ics, S.ics * w1s, bad);
        if (redo) {              // (the partial row just written is written again)
            if (tid == 0) atomic_add_agent(a.split_events + 0, 1);
            attempt += 1;
            sg -= 1;
            continue;
        }
        attempt = 0;
        CH_STAMP(3);
        CH_WGSTAMP(1 + (sg - sg0 < 2 ? sg - sg0 : 1));
    }
}

// Largest |observation| of every task's rows -> obs_absmax[task] (as the bits of a non-negative float: ordered like unsigned
// integers, NaN above everything).  Run once when a slab arrives (upload, end of a collection, device rollout): the FP16 split
// of the pass kernels keeps the observations near 1 with the power of two this yields (k_pass: W.xs).  Grid (tasks, slices);
// the table is zeroed in front of the launch.  HBM-bound: 4 O bytes per row, once per slab.
struct ObsRangeArgs {
    const float* obs;
    const int* task_row_offsets;
    unsigned* absmax;       // [tasks]
    int O;
};
__global__ void __launch_bounds__(256) k_obs_range(ObsRangeArgs a) {
    const int task = blockIdx.y, tid = threadIdx.x;
    const long long r0 = a.task_row_offsets[task], r1 = a.task_row_offsets[task + 1];
    const long long n = (r1 - r0) * a.O, per = (n + gridDim.x - 1) / gridDim.x;
    const long long b = per * blockIdx.x, e = b + per < n ? b + per : n;
    const float* src = a.obs + r0 * a.O;
    // four loads in flight per thread and trip; ONE atomic per workgroup (the waves' maxima meet in LDS first: a task's address
    // takes gridDim.x atomics, not four times that -- they serialise in L2 at a few hundred nanoseconds each)
    unsigned m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    long long i = b + tid;
    for (; i + 768 < e; i += 1024) {
        const unsigned u0 = __builtin_bit_cast(unsigned, src[i]) & 0x7FFFFFFFu, u1 = __builtin_bit_cast(unsigned, src[i + 256]) & 0x7FFFFFFFu;
        const unsigned u2 = __builtin_bit_cast(unsigned, src[i + 512]) & 0x7FFFFFFFu, u3 = __builtin_bit_cast(unsigned, src[i + 768]) & 0x7FFFFFFFu;
        m0 = u0 > m0 ? u0 : m0;
        m1 = u1 > m1 ? u1 : m1;
        m2 = u2 > m2 ? u2 : m2;
        m3 = u3 > m3 ? u3 : m3;
    }
    for (; i < e; i += 256) {
        const unsigned u = __builtin_bit_cast(unsigned, src[i]) & 0x7FFFFFFFu;
        m0 = u > m0 ? u : m0;
    }
    unsigned m = m0 > m1 ? m0 : m1;
    m = m > m2 ? m : m2;
    m = m > m3 ? m : m3;
    float mf = __builtin_bit_cast(float, m);      // (compared as integers below: the bit patterns of non-negative floats are ordered)
#pragma unroll
    for (int x = 32; x >= 1; x >>= 1) {
        const unsigned o = __builtin_bit_cast(unsigned, shfl_xor_f32(mf, x));
        m = o > m ? o : m;
        mf = __builtin_bit_cast(float, m);
    }
    PROMP_SMEM_DECL;
    unsigned* sm = (unsigned*)PROMP_SMEM_PTR;
    if ((tid & 63) == 0) sm[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        unsigned t = sm[0] > sm[1] ? sm[0] : sm[1], u = sm[2] > sm[3] ? sm[2] : sm[3];
        t = t > u ? t : u;
        if (t != 0) atomic_max_agent(a.absmax + task, t);
    }
}
