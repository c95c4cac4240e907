// promp_kernels_chain.h -- per-task Gaussian-MLP policy passes for hidden widths <= 64 (reference rows a8-a13).
//
//   k_chain_hvp  : out = -H v + kl_weight * grad KL, H = Hessian of the inner objective      (K12, K13)
// optionally followed, inside the same launch, by the fixed-order sum of a task's partial rows and the update that
// consumes it (multiplier update), done by whichever workgroup of the task finishes last.
// (The first-order pass k_pass, promp_kernels_pass.h, shares this file's segment table, primal cache and reductions.)
//
// Arithmetic follows oracle/promp.py (which restates meta_algos/pro_mp.py:59-155, meta_algos/base.py:192-215,
// policies/networks/mlp.py:65-119, policies/distributions/diagonal_gaussian.py:16-109 of the reference).
//
// Register-chained layers.  v_mfma_f32_16x16x4_f32 computes D[m][n] += sum_k A[m][k] B[k][n]; lane (i16, kk) feeds
// A[i16][kk] and B[kk][i16] and receives D[4 kk + r][i16], r = 0..3.  A wave owns a tile of 16 samples and keeps its
// activations TRANSPOSED: samples along n (= i16), units along m.  The four registers a lane receives are units
// 16 c + 4 kk + r of sample i16 -- exactly what lane (i16, kk) must feed as B[k][n] to the next layer if the
// contraction visits the units in the order k-step (c, r) <-> unit 16 c + 4 kk + r.  So H^T_next = W^T H^T runs from
// accumulator registers to operand registers with no LDS round trip, forward (weights as A[m = out unit][k = in unit])
// and backward (A[m = in unit][k = out unit]) alike; the tanh / cotangent epilogues are plain per-register VALU work.
// The weights are staged once per task into LDS in fragment order (each lane's four k-steps contiguous: one
// ds_read_b128 per 4 MFMAs, no address arithmetic); the backward pass reads the same copy with 4-byte reads (the
// lane-group rows are padded to 68 floats so that both access patterns are bank-conflict free).
//
// Only the weight gradients contract over SAMPLES and need both operands with the unit along i16: those operands
// go through a per-wave [16][68] LDS tile (written as 4 ds_write_b128 per 64 units, read as 4-byte columns); the
// contraction visits the samples in the order k-step t <-> sample 4 kk + t.  The observation tile is read from
// global memory in both orientations.
//
// Work split.  A 16-sample tile is the unit; the NW waves of a workgroup walk a segment (tiles [tile0, tile0+n) of
// one task) round-robin, so a segment costs ceil(n / NW) rounds.  The host cuts the global list of rounds into equal
// shares, one per CU; a share that straddles a task boundary becomes two segments that the workgroup walks one after
// the other (restaging the parameters in between).  Every (workgroup, segment) writes one partial row.
#pragma once
#include "promp_device.h"

#define PROMP_PASS_TPLANE 512    // words per plane of a [16 samples][64 units] tile of 16-bit halves
#define PROMP_PASS_XPLANE 256    // [16][32 observation slots]
#define PROMP_PASS_DPLANE 160    // [16][16 action slots], rows of 8 words, 32 words of padding between samples 7 and 8
#define PROMP_PARTIAL_EXTRA 4   // loss, kl, 2 spare
#define PROMP_CH_TS 68          // row stride of a [16 samples][<= 64 units] transpose tile
#define PROMP_CH_DS 20          // row stride of the [16 samples][16 action slots] cotangent tile
#define PROMP_CH_ROW 68         // floats per lane-group row of a hidden_1 fragment block (64 + 4 pad)
#define PROMP_CH_BLK (4 * PROMP_CH_ROW)

enum { LOSS_RATIO = 0, LOSS_CLIP = 1, LOSS_LOGLIK = 2, LOSS_KL = 3 };   // LOSS_KL: mean KL(old || new) itself (TRPO constraint)

// Which entry of the workgroup -> segments table a workgroup of k_pass / k_chain_hvp takes.  The table is ordered by task, and the
// dispatcher deals consecutive workgroups out over the eight XCDs: with the identity (0) the ~6 workgroups of a task sit on different
// XCDs and each of the eight L2s fetches every task's parameters; with xcd_item (1) an XCD takes a contiguous eighth of the table --
// five tasks at config 3 -- and a task's parameters, direction and partial rows stay in one L2.  Measured at config 3 (round 6, two
// A/B pairs on one box): k_pass 50.0 -> 50.5 us, k_chain_hvp 71.8 -> 72.5 us, forward-only 27.3 -> 26.7 us, step 1.093 -> 1.091 ms:
// nothing -- these kernels read their parameters once per segment (24 KB, 7 % of a segment with the loads of all lanes in flight),
// unlike the cooperative kernels, which re-read them every round and do use the mapping.  The identity stays.
#ifndef PROMP_PASS_XCD
#define PROMP_PASS_XCD 0
#endif
struct WorkItem {
    int task, row_begin, row_end, pad;
};

// Primal cache.  The R-operator pass at theta on a step's slab recomputes what the gradient pass at the SAME theta on the
// SAME slab computed moments earlier: the hidden activations and the means.  Both kernels are bound by the matrix pipe
// while HBM idles (2 % of its bandwidth), so the gradient pass writes them out and the R-operator pass reads them back,
// together with the hidden_0 cotangent before its tanh derivative (the product W2 dZ2^T, which the tangent of that
// derivative needs): 164 of its 688 MFMAs per tile and all 32 tanh per lane disappear for 4 (2 H1 + H2 + 8) bytes per row
// each way.  One block per 16-row tile, in the R-operator pass's operand order (a lane's four units contiguous):
//     [c < NC1][sample 16][unit 16]  hidden_0 activations     [c < NC2][sample 16][unit 16]  hidden_1 activations
//     [sample 16][action 8]          means
//     [c < NC1][sample 16][unit 16]  W2 dZ2^T
// The block of tile t of a task starts at "row" row0(task) + 16 t + 16 task: a task's last tile may be partial, the 16
// spare rows per task keep the next task's first block clear of it without a tile-offset table.
PROMP_CX int chain_cache_row(int H1, int H2) { return 2 * H1 + H2 + 8; }     // floats per row

struct ChainSeg {
    int task, tile0, ntiles, pad;   // 16-row tiles [tile0, tile0 + ntiles) of the task; slot = index of the segment
};

// What the last-arriving workgroup of a task does with the task's summed partial row g (fixed slot order):
//   RED_STEP  : next[i] = cur[i] - alpha * g                  ; scal[i] = {loss, kl}     (inner SGD step)
//   RED_OUTER : lam[i] = g ; v[i] = alpha * g                 ; scal[i] = {loss, kl}
//   RED_HVP   : lam[i] += g ; v[i] = alpha * lam[i]           ; scal[i] = {-, kl}
//   RED_PLAIN : lam[i] = g                                    ; scal
//   RED_SCAL  : scal only (forward-only passes write no gradient)
enum { RED_STEP = 0, RED_OUTER = 1, RED_HVP = 2, RED_PLAIN = 3, RED_SCAL = 4 };

struct PassArgs {
    const float* obs;           // [rows][O]
    const float* obs_absmax;    // [tasks]: the largest |observation| of each task's rows (k_obs_range, when the slab arrived); may be NULL
    const float* act;           // [rows][A]
    const float* adv;           // [rows]
    const float* old_mean;      // [rows][A]
    const float* old_log_std;   // [rows][A] or [tasks][A]
    int ls_per_row;
    const int* task_row_offsets;  // [tasks+1]
    const WorkItem* work;         // cooperative (wide) kernels: one row range per workgroup
    const ChainSeg* segs;         // k_pass, k_chain_hvp
    const int* wg_seg_offsets;    // [grid+1]
    const float* theta;           // [Theta] or [tasks][Theta]
    long long theta_task_stride;  // 0 => shared
    const float* vdir;            // hvp: [tasks][Theta]
    float* partials;              // [slots][partial_stride]
    int partial_stride;
    int O, A;
    int loss_kind;
    float clip_eps;
    int clip_log_std;
    float min_log_std;
    float kl_weight;
    // fused per-task reduction (chain kernels)
    int* task_counters;             // [tasks]; zero between launches (the last arriver resets its task's counter)
    const int* task_slot_offsets;   // [tasks+1]
    int red_mode;
    int fuse_reduce;                // chain kernels: the last-arriving workgroup of a task sums its partial rows (else k_reduce_task follows)
    const float* step_sizes;        // [Theta]
    const float* cur;               // RED_STEP: [Theta] or [tasks][Theta]
    long long cur_task_stride;
    float* next;                    // [tasks][Theta]
    float* lam;                     // [tasks][Theta]
    float* v;                       // [tasks][Theta]
    float* scal;                    // [tasks][2]
    float* row_tan;                 // k_chain_hvp, optional [rows]: R'{log pi} of every row = dlogpi_row . (-v)  (DiCE coupling)
    float* hcache;                  // primal cache of the step (k_pass<STORE> writes it, k_chain_hvp<CACHED> reads it), see chain_cache_row
    unsigned long long* dbg;        // optional cycle stamps (developer tooling), else NULL
    int* split_events;              // [2] FP16 split: segments walked again because a cotangent left the format, by k_pass [0] and by
                                    // k_chain_hvp [1] (promp_split_events); counted when it happens
    // BF16-pipe cooperative kernels (promp_kernels_wide_bf16.h): the parameters' (and the direction's) hidden kernels as pre-split
    // BF16 planes in fragment order, written by k_wb_planes right before the pass
    const float* vdir_absmax;       // [tasks]: the largest |entry| of each task's direction (k_vec_absmax), the FP16 split's scale for it
    const unsigned* wb_theta_planes;
    const unsigned* wb_v_planes;
    long long wb_plane_stride;      // words per task; 0 => shared parameters
};

// Layer 2 of the R-operator pass (tangent, and the primal product where it is not read from the cache) runs on the BF16 matrix pipe
// (error-compensated 3-way split of both operands, 6 of the 9 products, float32 accumulation: at least as accurate as the FP32
// MFMA chain; guarded by tests/test_gpu_parity.py::test_split_gemm_accuracy_guard).

struct ChainLds {
    int w1, w2, w3, w3b, b1, b2, b3, dist;   // inside one network block
    int net_stride;
    int planes, plane_stride;                // BF16 planes of the two networks' hidden_1 kernels, behind both network blocks
    int bplanes, bplane_stride;              // (bwdp) the same kernels' planes in the orientation of the backward product
    int flag;                                // one int: "this workgroup arrived last"
    int wave0, wave_stride, tb0, tb1, db0, db1;
    int tp, tq, tah, xt, dm0, dm1;           // (bwdp) the cached instance's plane tiles (chain_layout)
    int vmx;                                 // one float per wave: the largest |direction entry| its staging share holds (FP16 split)
    int total;
};
#define PROMP_CH_TPL 512         // words per plane of a [16 samples][64 units] bf16 tile (B operand of the hidden_1 kernel gradient)
#define PROMP_CH_APL 256         // words per plane of a [16 samples][32 units] bf16 half tile (its A operand, one 32-unit block at a time)

// dist block of the theta network: 6 x 8 floats
enum { CH_LS = 0, CH_ES = 8, CH_SN2 = 16, CH_LMASK = 24, CH_VLS = 32, CH_RDEN = 40 };

// bwdp (the cache-reading R-operator pass): the backward product W2 qZ2^T + (-vW2) dZ2^T runs on the BF16 pipe too and needs both
// networks' hidden_1 planes in the second orientation (48 KB at 64/64).  LDS pays for them with what that pass no longer reads:
// the float32 hidden_1 fragments of both networks (every product with W2 is on planes now), theta's hidden_0 fragments (h1 comes
// from the cache), and the padding lanes of the output kernel's fragments (stored by action instead of by action slot).
PROMP_CX ChainLds chain_layout(int NC1, int NC2, int nwaves, bool hvp, int NP, bool bwdp = false) {
    ChainLds L{};
    int o = 4;                    // [0, 4): flag
    L.flag = 0;
    int n = 0;
    if (!bwdp) {
        L.w1 = n;  n += NC1 * 512;                  // [c][t4][lane][4]: W1[4 (4 t4 + r) + kk][16 c + i16], obs padded to 32
        L.w2 = n;  n += NC2 * NC1 * PROMP_CH_BLK;   // [c2][c1][kk][i16][r]: W2[16 c1 + 4 kk + r][16 c2 + i16]
    }
    L.w3 = n;  n += bwdp ? PROMP_NT * (NC2 / 2) * 256 : NC2 * 256;
                                                    // [c][lane][r]: W3[16 c + 4 kk + r][a(i16)], a(4 ko + ro) = 2 ko + ro (ro < 2)
                                                    // bwdp (round 6): the split's planes [term][P][lane] x 16 B, k_pass's output fragments:
                                                    // W3[16 (2P + e / 4) + 4 kk + e % 4][action of row i16]
    L.w3b = n; n += NC2 * 128;                      // [c][lane][ro]: W3[16 c + i16][2 kk + ro]
    L.b1 = n;  n += 16 * NC1;
    L.b2 = n;  n += 16 * NC2;
    L.b3 = n;  n += 8;
    L.dist = n; n += 48;
    L.net_stride = n;
    o += (hvp ? 2 : 1) * n;
    if (bwdp) {                   // the direction's hidden_0 kernel alone: (network block 0) + L.w1 + net_stride lands here.  Round 6:
        L.w1 = (o - 4) - n;       // as the split's planes [term][c][lane] x 16 B, k_pass's layer-1 fragments: W1[obs 8 kk + e][16 c + i16]
        L.w2 = 0;
        o += PROMP_NT * NC1 * 256;
    }
    // BF16 planes of the hidden_1 kernel for v_mfma_f32_16x16x32_bf16: [term 3][c2][pair of input blocks][lane] x 8 bf16 (16 B):
    // lane (i16, kk) of chunk (c2, P): W2[16 (2P) + 4 kk + r][16 c2 + i16], r = 0..3, then W2[16 (2P + 1) + 4 kk + r][.]
    // (behind BOTH network blocks: inside them they would push the second network's float32 fragments past the 64 KB that
    // a ds_read immediate offset reaches, and every such read would cost an address add: +4.5 us per launch, measured)
    L.planes = o;
    L.plane_stride = PROMP_NT * NC2 * (NC1 / 2) * 256;
    o += hvp ? 2 * L.plane_stride : 0;
    // bwdp: [term 3][c1][pair of hidden_1 OUTPUT blocks P][lane] x 8 bf16: lane (i16, kk) of chunk (c1, P):
    // W2[16 c1 + i16][16 (2P) + 4 kk + r], r = 0..3, then W2[16 c1 + i16][16 (2P + 1) + 4 kk + r]
    L.bplanes = o;
    L.bplane_stride = PROMP_NT * NC1 * (NC2 / 2) * 256;
    o += bwdp ? 2 * L.bplane_stride : 0;
    L.wave0 = o;
    int q = 0;
    if (bwdp) {
        // round 6: every product of the cached instance runs on the split's planes (k_pass's tiles): two full plane tiles (tangent /
        // primal hidden_1 activations for the output-kernel gradient; then the two operands of each stage of the hidden_1 kernel
        // gradient; then qZ1), the observations, the two cotangents of the mean.  No float32 tile is left.
        L.tb0 = L.tb1 = L.db0 = L.db1 = 0;
        L.tp = q;  q += PROMP_NT * PROMP_CH_TPL;
        L.tq = q;  q += PROMP_NT * PROMP_CH_TPL;
        L.tah = 0;
        L.xt = q;  q += PROMP_NT * PROMP_PASS_XPLANE;
        L.dm0 = q; q += PROMP_NT * PROMP_PASS_DPLANE;
        L.dm1 = q; q += PROMP_NT * PROMP_PASS_DPLANE;
    } else {
        L.tb0 = q; q += 16 * PROMP_CH_TS;
        L.tb1 = q; q += 16 * PROMP_CH_TS;
        L.db0 = q; q += 16 * PROMP_CH_DS;
        L.db1 = q; q += hvp ? 16 * PROMP_CH_DS : 0;
    }
    L.wave_stride = q;
    o += nwaves * q;
    {   // end-of-segment: one slab of [NP + 2] floats per wave, from offset 4 (aliases everything else)
        const int need = 4 + nwaves * ((NP + 2 + 3) & ~3);
        if (o < need) o = need;
    }
    L.vmx = o;                    // behind everything, the slabs included: written at the head of a segment, before its first barrier
    o += 4 * ((nwaves + 3) / 4);
    L.total = o;
    return L;
}

// developer tooling: cycle stamps of workgroup 0 / thread 0 (compiled in only with -DPROMP_DEV_STAMPS, tools/phase_timing.py)
#ifdef PROMP_DEV_STAMPS
#define CH_STAMP(i) do { if (a.dbg != nullptr && blockIdx.x == 0 && tid == 0) a.dbg[(i)] = promp_clock(); } while (0)
#define CH_TSTAMP(j) CH_STAMP(8 + 16 * (tix < 3 ? tix : 3) + (j))
// every workgroup: wall clock (100 MHz, chip-wide) when it starts / ends, and its XCC id
#define CH_WGSTAMP(j) do { if (a.dbg != nullptr && tid == 0) a.dbg[256 + 4 * blockIdx.x + (j)] = promp_wall_clock(); } while (0)
#define PROMP_STAMPS_ON 1
#else
#define CH_STAMP(i) do { } while (0)
#define CH_TSTAMP(j) do { } while (0)
#define CH_WGSTAMP(j) do { } while (0)
#define PROMP_STAMPS_ON 0
#endif

PROMP_DEV f32x4 splat4(float v) {
    f32x4 z;
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = v;
    return z;
}
PROMP_DEV f32x4 lds4(const float* p) { return *(const f32x4*)p; }
PROMP_DEV f32x2 lds2(const float* p) { return *(const f32x2*)p; }
PROMP_DEV void sts4(float* p, f32x4 v) { *(f32x4*)p = v; }
PROMP_DEV void sts2(float* p, f32x2 v) { *(f32x2*)p = v; }

// ---- BF16 operand planes through LDS (shared with k_pass, promp_kernels_pass.h) -------------------------------------------
// Weight gradients contract over samples and need both operands with the unit along the lane index.  An operand's three BF16
// planes go through a per-wave LDS tile of 8-byte chunks (4 units of one sample, written by the chain side with ds_write_b64)
// and come back through ds_read_b64_tr_b16, the 4 x 16 transpose read of gfx950: two reads give a lane the eight samples of
// its unit, the products run on v_mfma_f32_32x32x16_bf16 (K = 16 samples).
// chunk (sample s of 16, unit chunk q = unit / 4) of a transposed tile -> its 8-byte slot.  For the writer (16 lanes = 16
// samples of one chunk column: ds_write_b64, banks mod 32) the low four bits are a bijection of s; for the transpose read of a
// 32-unit block (32 lanes = 4 samples x 8 chunks, banks mod 64) the low five bits are a bijection of (s & 3, q & 7).
PROMP_CX int pass_slot(int s, int q) { return 32 * (4 * (q >> 3) + (q & 3)) + 16 * ((q >> 2) & 1) + 4 * (s & 3) + ((s >> 2) ^ (q & 3)); }

// per-lane addresses (in words) of the transposed tiles (k_pass; k_chain_hvp's cached instance)
struct PassTileAddr {
    int wr, xw0, xw1, rd32_0, rd32_1, rd16_0, rd16_1, dmw, dr0, dr1;
};
PROMP_DEV PassTileAddr pass_tile_addr(int lane) {
    PassTileAddr T;
    const int i16 = lane & 15, kk = lane >> 4, p16 = lane & 15, g32 = (lane >> 4) & 1, kh = lane >> 5;
    // where this lane writes (chain side: sample i16, chunk 4 c + kk) ...
    T.wr = 2 * pass_slot(i16, kk);
    T.xw0 = 2 * pass_slot(i16, 2 * kk);
    T.xw1 = 2 * pass_slot(i16, 2 * kk + 1);
    // ... and reads: 32-unit blocks (v_mfma_f32_32x32x16 operands): samples 8 kh + 4 t + p16 / 4, chunk 8 b + 4 g32 + p16 % 4
    T.rd32_0 = 2 * pass_slot(8 * kh + (p16 >> 2), 4 * g32 + (p16 & 3));
    T.rd32_1 = 2 * pass_slot(8 * kh + 4 + (p16 >> 2), 4 * g32 + (p16 & 3));
    // 16-unit blocks (A operand of the output-kernel gradient on v_mfma_f32_16x16x32): samples 8 (kk & 1) + 4 t + p16 / 4
    // (the k-slots of the lane groups kk >= 2 meet zeros on the B side; they read the same finite data as kk - 2)
    T.rd16_0 = 2 * pass_slot(8 * (kk & 1) + (p16 >> 2), p16 & 3);
    T.rd16_1 = 2 * pass_slot(8 * (kk & 1) + 4 + (p16 >> 2), p16 & 3);
    // cotangent-of-the-mean tile [16 samples][16 action slots]: rows of 8 words, 32 words of padding after sample 7
    T.dmw = 8 * i16 + 32 * (i16 >> 3) + kk;
    T.dr0 = (kk < 2) ? 8 * (8 * kk + (p16 >> 2)) + 32 * kk + 2 * (p16 & 3) : 4;     // kk >= 2: a chunk of zeros (actions 8..11 of sample 0)
    T.dr1 = (kk < 2) ? 8 * (8 * kk + 4 + (p16 >> 2)) + 32 * kk + 2 * (p16 & 3) : 4;
    return T;
}

PROMP_DEV void sts_w2(float* p, unsigned a, unsigned b) {
    u32x2 v;
    v[0] = a;
    v[1] = b;
    *(u32x2*)p = v;
}
PROMP_DEV void sts_w4(float* p, u32x4 v) { *(u32x4*)p = v; }
PROMP_DEV u32x4 join_w2(u32x2 lo, u32x2 hi) {
    u32x4 v;
    v[0] = lo[0];
    v[1] = lo[1];
    v[2] = hi[0];
    v[3] = hi[1];
    return v;
}
// a lane's eight k-slots of one input chunk (blocks 2P, 2P + 1) -> the planes of the split (NT = 2: FP16, 3: BF16)
template <int NT>
PROMP_DEV void pass_split8(const f32x4& lo, const f32x4& hi, u32x4 (&pl)[NT]) {
    unsigned w0[NT], w1[NT], w2[NT], w3[NT];
    split_pair<NT>(lo[0], lo[1], w0);
    split_pair<NT>(lo[2], lo[3], w1);
    split_pair<NT>(hi[0], hi[1], w2);
    split_pair<NT>(hi[2], hi[3], w3);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        pl[t][0] = w0[t];
        pl[t][1] = w1[t];
        pl[t][2] = w2[t];
        pl[t][3] = w3[t];
    }
}
// a chunk pair (blocks 2P, 2P + 1 of sample i16) of the planes -> a transposed tile
template <int NT>
PROMP_DEV void pass_store_planes(float* tile, int plane_words, int off, const u32x4 (&pl)[NT]) {
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        sts_w2(tile + tt * plane_words + off, pl[tt][0], pl[tt][1]);
        sts_w2(tile + tt * plane_words + off + 32, pl[tt][2], pl[tt][3]);
    }
}
// the planes of a 32-unit (or 16-unit) block as a lane's eight samples: two transpose reads each
template <int NT>
PROMP_DEV void pass_read_tr(u32x4 (&fr)[NT], const float* tile, int plane_words, int rd0, int rd1) {
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) fr[tt] = join_w2(lds_tr16(tile + tt * plane_words + rd0), lds_tr16(tile + tt * plane_words + rd1));
}
// The task's observations are multiplied by 2^-sx, the hidden_0 kernels by 2^sx: the largest observation (k_obs_range) goes to
// [1, 2).  Both sides of the product then sit where the split is exact to 2^-22 for the networks the reference builds (Xavier
// kernels, observations of size ~10: the kernel's entries come to ~1 as well -- unshifted most of them are below 2^-3, where a
// value's low term is subnormal); a network that reads observations of size 2^15 has a hidden_0 kernel of size 2^-12 or saturated
// units, and the other way round.  sx <= 12 keeps the kernel's side inside the format for entries up to 5;
// |W1| max |obs| <= 2^17 is the limit of the fused kernels' FP16 split (beyond it: infinities, loudly).
PROMP_DEV int obs_shift(const float* obs_absmax, int task) {
    if (PROMP_NT != 2 || obs_absmax == nullptr) return 0;
    const float mx = obs_absmax[task];
    const int sx = (mx > 0.f && mx < 3.0e38f) ? -scale_exp(mx, 0) : 0;
    return sx < -24 ? -24 : sx > 12 ? 12 : sx;
}
// the largest |value| of a wave, in every lane (once per segment: scales of the FP16 split's operands)
PROMP_DEV float wave_absmax_f32(float v) {
    v = fabsf(v);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor_f32(v, m));
    return v;
}


// The networks of a segment (theta, and the direction, negated) -> fragment order in LDS, plus the BF16 planes of both
// hidden_1 kernels.  A wave stages whole blocks of 64 four-float fragments (the block is wave-uniform, so a fragment's source
// indices are a lane-constant pattern shifted by scalars: no per-element index arithmetic, one ds_write_b128 per fragment);
// the hidden_1 blocks come in pairs (input blocks 2P, 2P + 1 of one output block): the pair IS a lane's eight k-slots of the
// BF16 instruction, so the three planes are split from the registers that were just loaded.  All global loads of a wave
// are issued before its first store (one round trip to L2 / memory: the parameters were written by another CU's reduction
// moments ago): nothing in the load phase USES a loaded value or branches (otherwise the blocks' round trips add up: 13 k cycles
// instead of 3 k, measured).  Padding reads a clamped valid element; signs and the zero masks of the padding are applied in the
// store phase.
//   blocks, per network:  [0, 2 NC1)                       hidden_0: (c, t4)   W1[4 (4 t4 + r) + kk][16 c + i16], obs padded to 32
//                         [.., + NC2 NC1 / 2)              hidden_1 pairs: (c2, P)  W2[16 (2P + h) + 4 kk + r][16 c2 + i16], h = 0, 1
//                         [.., + NC2)                      output: (c)  W3[16 c + 4 kk + r][a(i16)], a(4 ko + ro) = 2 ko + ro (ro < 2)
//                                                          and its [c][lane][ro] copy W3[16 c + i16][2 kk + ro]; the last block
//                                                          also carries the biases
// `mid` runs between the two phases (all of this function's loads issued, none used, LDS untouched so far): the caller requests its
// first tile's inputs there, then the workgroup joins -- those requests go to memory (the observation slab, the primal cache) and
// would, issued first, hold back the parameters behind them (a wave's loads return in order; the parameters come from L2).
// FP16 split: the direction is staged multiplied by -2^kv, the power of two that brings its largest entry (over the whole vector:
// every wave's share, exchanged through `vmx` across the barrier behind `mid`) to [2^vt, 2^(vt+1)) -- the hidden_0 block weighed by
// w1w = 2^sx, the inverse of the observations' scale: what it contributes to a tangent is its product with an observation, and the
// blocks are to be compared by what they contribute.  The R-operator is linear in the
// direction, so every tangent the kernel computes simply carries 2^kv (returned; undone on the accumulators at the end of the
// segment).  The three-term BF16 split has no range and returns 1.
template <int NC1, int NC2, int NW, bool BWDP, typename Mid>
PROMP_DEV float chain_stage_nets(float* sm, float* vmx, const float* src0, const float* src1, int O, int A, int tid, int vt, float w1w, Mid&& mid) {
    constexpr int H1 = 16 * NC1, H2 = 16 * NC2, NT = 64 * NW;
    constexpr ChainLds L = chain_layout(NC1, NC2, 1, true, 0, BWDP);
    constexpr int NB1 = 2 * NC1, NB2 = NC2 * (NC1 / 2), NB3 = NC2, NB4 = BWDP ? NC1 * (NC2 / 2) : 0;
    constexpr int IT1 = (NB1 + NW - 1) / NW, IT2 = (NB2 + NW - 1) / NW, IT3 = (NB3 + NW - 1) / NW, IT4 = (NB4 + NW - 1) / NW;
    constexpr int N5 = H1 + H2 + 8, IS = (N5 + NT - 1) / NT;
    // An opaque copy of the thread index: the lane-constant index arithmetic below is otherwise hoisted out of the segment loop,
    // kept alive across the tile loop -- which has no register to spare -- and spilled.  A scratch reload in the middle of the load
    // phase waits for every load issued before it (memory returns a wave's loads in order), and the phase becomes two round trips
    // to memory: 9.4 k instead of 5.4 k cycles until the last load is issued, k_chain_hvp 100.2 -> 96.4 us per launch (measured).
    tid += opaque_zero();
    const int lane = tid & 63, i16 = lane & 15, kk = lane >> 4, w = wave_uniform(tid >> 6);
    const int ob1 = O * H1, oW2 = ob1 + H1, ob2 = oW2 + H1 * H2, oW3 = ob2 + H2, ob3 = oW3 + H2 * A;
    // output block: the action of row i16.  Fragment order by action SLOT (slot 4 ko + ro <-> action 2 ko + ro, ro < 2; the other
    // lanes hold zeros), or -- BWDP -- by action: lanes i16 and i16 + 8 both stage action i16 & 7 (same values, same addresses)
    const int mm = i16, ro3 = mm & 3, aa3 = BWDP ? (mm & 7) : 2 * (mm >> 2) + ro3;
    const bool ok3 = (BWDP || ro3 < 2) && aa3 < A;
    // One loop per block kind with a compile-time trip count; a wave whose block index runs past the end of a kind loads and
    // stores the kind's last block a second time (same values, same addresses): no branch anywhere (loads inside a block-kind
    // branch make the compiler wait, at the head of the next branch, for loads it believes may still target the registers it
    // reuses; a guarded store invites it to sink the block's loads into the guard).
    // (BWDP, round 6: the direction's hidden_0 kernel and both output kernels as the split's planes in k_pass's fragment order --
    //  NC1 blocks of eight observation slots per lane, NC2 / 2 blocks of eight units per lane -- instead of float32 fragments)
    constexpr int NB1P = BWDP ? NC1 : 0, NB3P = BWDP ? NC2 / 2 : 0;
    constexpr int IT1P = (NB1P + NW - 1) / NW, IT3P = (NB3P + NW - 1) / NW;
    const int aro = i16 & 3, aact = 2 * (i16 >> 2) + aro;        // k_pass's output rows: action 2 (m / 4) + m % 4 for m % 4 < 2
    const bool aok = aro < 2 && aact < A;
    float x1[2][IT1][4], x2[2][IT2][8], x3[2][IT3][4], y3[2][IT3][2], x4[2][IT4 ? IT4 : 1][8], z[2][IS];
    float x1p[IT1P ? IT1P : 1][8], x3p[2][IT3P ? IT3P : 1][8];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const float* src = n ? src1 : src0;
        if (!BWDP) {
#pragma unroll
            for (int it = 0; it < IT1; ++it) {
                const int bj = w + it * NW, b = bj < NB1 ? bj : NB1 - 1, c = b >> 1, t4 = b & 1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 4 * (4 * t4 + r) + kk;
                    x1[n][it][r] = src[(o < O ? o : O - 1) * H1 + 16 * c + i16];
                }
            }
        } else if (n == 1) {
#pragma unroll
            for (int it = 0; it < IT1P; ++it) {
                const int bj = w + it * NW, b = bj < NB1P ? bj : NB1P - 1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int o = 8 * kk + e;
                    x1p[it][e] = src[(o < O ? o : 0) * H1 + 16 * b + i16];
                }
            }
        }
        if (BWDP) {
#pragma unroll
            for (int it = 0; it < IT3P; ++it) {
                const int bj = w + it * NW, P = bj < NB3P ? bj : NB3P - 1;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    x3p[n][it][e] = src[oW3 + (16 * (2 * P + (e >> 2)) + 4 * kk + (e & 3)) * A + (aok ? aact : 0)];
            }
        }
#pragma unroll
        for (int it = 0; it < IT2; ++it) {
            const int bj = w + it * NW, b = bj < NB2 ? bj : NB2 - 1, c2 = b / (NC1 / 2), P = b - c2 * (NC1 / 2);
#pragma unroll
            for (int e = 0; e < 8; ++e) x2[n][it][e] = src[oW2 + (16 * (2 * P + (e >> 2)) + 4 * kk + (e & 3)) * H2 + 16 * c2 + i16];
        }
#pragma unroll
        for (int it = 0; it < IT4; ++it) {
            const int bj = w + it * NW, b = bj < NB4 ? bj : NB4 - 1, c1 = b / (NC2 / 2), P = b - c1 * (NC2 / 2);
#pragma unroll
            for (int e = 0; e < 8; ++e) x4[n][it][e] = src[oW2 + (16 * c1 + i16) * H2 + 16 * (2 * P + (e >> 2)) + 4 * kk + (e & 3)];
        }
#pragma unroll
        for (int it = 0; it < IT3; ++it) {
            const int bj = w + it * NW, c = bj < NB3 ? bj : NB3 - 1;
            if (!BWDP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) x3[n][it][r] = src[oW3 + (16 * c + 4 * kk + r) * A + (ok3 ? aa3 : 0)];
            }
#pragma unroll
            for (int ro = 0; ro < 2; ++ro) {
                const int aa = 2 * kk + ro;
                y3[n][it][ro] = src[oW3 + (16 * c + i16) * A + (aa < A ? aa : 0)];
            }
        }
    }
#pragma unroll
    for (int it = 0; it < IS; ++it) {
        const int dj = tid + it * NT, d = dj < N5 ? dj : N5 - 1;
        const int aa = d - H1 - H2;
        const int idx = d < H1 ? ob1 + d : d < H1 + H2 ? ob2 + d - H1 : ob3 + (aa < A ? aa : 0);
        z[0][it] = src0[idx];
        z[1][it] = src1[idx];
    }
    sched_fence();       // every load is issued before the first store (the scheduler would interleave them in batches: several round trips)
    mid();               // (the caller's requests for its first tile: behind the parameters', in front of the barrier)
    if (PROMP_NT == 2) {
        float m = 0.f;
#pragma unroll
        for (int it = 0; it < (BWDP ? 0 : IT1); ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, w1w * fabsf(x1[1][it][r]));
#pragma unroll
        for (int it = 0; it < IT1P; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, w1w * fabsf(x1p[it][e]));
#pragma unroll
        for (int it = 0; it < IT2; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(x2[1][it][e]));
#pragma unroll
        for (int it = 0; it < (BWDP ? 0 : IT3); ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(x3[1][it][r]));
#pragma unroll
        for (int it = 0; it < IT3P; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(x3p[1][it][e]));
#pragma unroll
        for (int it = 0; it < IS; ++it) m = fmaxf(m, fabsf(z[1][it]));
        m = wave_absmax_f32(m);
        if (lane == 0) vmx[w] = m;       // (the KERNEL's layout: behind its slabs; the one-wave layout of this function ends earlier)
    }
    __syncthreads();     // the previous segment is done with LDS; every wave's share of the direction's range has been written
    float vs = 1.f;
    if (PROMP_NT == 2) {
        float m = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) m = fmaxf(m, vmx[ww]);
        int k = (m > 0.f && m < 3.0e38f) ? scale_exp(m, vt) : 0;
        k = k < -100 ? -100 : k > 100 ? 100 : k;
        vs = pow2f(k);
    }
    sched_fence();
    float* net0 = sm + 4;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        float* nb = net0 + n * L.net_stride;
        const float sg = n ? -vs : 1.f;          // the direction is staged negated (and scaled: FP16 split)
        if (!BWDP) {
#pragma unroll
            for (int it = 0; it < IT1; ++it) {
                const int bj = w + it * NW, b = bj < NB1 ? bj : NB1 - 1, t4 = b & 1;
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = x1[n][it][r] * (4 * (4 * t4 + r) + kk < O ? sg : 0.f);
                sts4(nb + L.w1 + b * 256 + lane * 4, v);
            }
        } else if (n == 1) {      // the direction's hidden_0 planes (they also take the inverse of the observations' scale, w1w)
#pragma unroll
            for (int it = 0; it < IT1P; ++it) {
                const int bj = w + it * NW, b = bj < NB1P ? bj : NB1P - 1;
                f32x4 lo, hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = x1p[it][e] * (8 * kk + e < O ? sg * w1w : 0.f);
                    hi[e] = x1p[it][4 + e] * (8 * kk + 4 + e < O ? sg * w1w : 0.f);
                }
                u32x4 t[PROMP_NT];
                pass_split8(lo, hi, t);
#pragma unroll
                for (int sp = 0; sp < PROMP_NT; ++sp) sts_w4(nb + L.w1 + (sp * NB1P + b) * 256 + lane * 4, t[sp]);
            }
        }
        if (BWDP) {               // both output kernels' planes
#pragma unroll
            for (int it = 0; it < IT3P; ++it) {
                const int bj = w + it * NW, P = bj < NB3P ? bj : NB3P - 1;
                f32x4 lo, hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = x3p[n][it][e] * (aok ? sg : 0.f);
                    hi[e] = x3p[n][it][4 + e] * (aok ? sg : 0.f);
                }
                u32x4 t[PROMP_NT];
                pass_split8(lo, hi, t);
#pragma unroll
                for (int sp = 0; sp < PROMP_NT; ++sp) sts_w4(nb + L.w3 + (sp * NB3P + P) * 256 + lane * 4, t[sp]);
            }
        }
#pragma unroll
        for (int it = 0; it < IT2; ++it) {
            const int bj = w + it * NW, b = bj < NB2 ? bj : NB2 - 1, c2 = b / (NC1 / 2), P = b - c2 * (NC1 / 2);
            f32x4 lo, hi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lo[r] = x2[n][it][r] * sg;
                hi[r] = x2[n][it][4 + r] * sg;
            }
            // the pair is this lane's eight k-slots of the matrix instruction: the planes [term][c2][P][lane] x 8 halves
            u32x4 t[PROMP_NT];
            pass_split8(lo, hi, t);
            if (!BWDP) {
                sts4(nb + L.w2 + (c2 * NC1 + 2 * P) * PROMP_CH_BLK + kk * PROMP_CH_ROW + i16 * 4, lo);
                sts4(nb + L.w2 + (c2 * NC1 + 2 * P + 1) * PROMP_CH_BLK + kk * PROMP_CH_ROW + i16 * 4, hi);
            }
            float* pl = sm + L.planes + n * L.plane_stride;
#pragma unroll
            for (int sp = 0; sp < PROMP_NT; ++sp) sts_w4(pl + (((sp * NC2 + c2) * (NC1 / 2) + P) * 64 + lane) * 4, t[sp]);
        }
#pragma unroll
        for (int it = 0; it < IT4; ++it) {       // (BWDP) the planes of the backward product: [term][c1][P][lane] x 8 bf16
            const int bj = w + it * NW, b = bj < NB4 ? bj : NB4 - 1, c1 = b / (NC2 / 2), P = b - c1 * (NC2 / 2);
            f32x4 lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[e] = x4[n][it][e] * sg;
                hi[e] = x4[n][it][4 + e] * sg;
            }
            u32x4 t[PROMP_NT];
            pass_split8(lo, hi, t);
            float* pl = sm + L.bplanes + n * L.bplane_stride;
#pragma unroll
            for (int sp = 0; sp < PROMP_NT; ++sp) sts_w4(pl + (((sp * NC1 + c1) * (NC2 / 2) + P) * 64 + lane) * 4, t[sp]);
        }
#pragma unroll
        for (int it = 0; it < IT3; ++it) {
            const int cj = w + it * NW, c = cj < NB3 ? cj : NB3 - 1;
            f32x2 u;
            u[0] = y3[n][it][0] * (2 * kk < A ? sg : 0.f);
            u[1] = y3[n][it][1] * (2 * kk + 1 < A ? sg : 0.f);
            if (!BWDP) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = x3[n][it][r] * (ok3 ? sg : 0.f);
                sts4(nb + L.w3 + c * 256 + lane * 4, v);
            }
            sts2(nb + L.w3b + c * 128 + lane * 2, u);
        }
    }
#pragma unroll
    for (int it = 0; it < IS; ++it) {
        const int dj = tid + it * NT, d = dj < N5 ? dj : N5 - 1;
        const int off = d < H1 ? L.b1 + d : d < H1 + H2 ? L.b2 + d - H1 : L.b3 + d - H1 - H2;
        const float m = d < H1 + H2 || d - H1 - H2 < A ? 1.f : 0.f;
        net0[off] = z[0][it] * m;
        net0[L.net_stride + off] = z[1][it] * (-m * vs);
    }
    return vs;
}

// distribution constants of the theta network: s (clipped), exp(-s), exp(2 s), gradient mask, R{s}, 1 / (2 exp(2 s) + 1e-8).
// The two global reads are issued by chain_dist_load BEFORE the networks are staged (their round trip to L2 hides behind the
// staging loads instead of following them).
struct ChainDistRaw { float s, v; };
PROMP_DEV ChainDistRaw chain_dist_load(const float* th, const float* v, int oS, int A, int tid) {
    ChainDistRaw r;
    const int q = (tid < A) ? tid : 0;
    r.s = th[oS + q];
    r.v = (v != nullptr) ? v[oS + q] : 0.f;
    return r;
}
PROMP_DEV void chain_stage_dist(float* dist, ChainDistRaw raw, int A, int clip_log_std, float min_log_std, int tid, float vs = 1.f) {
    if (tid < 8) {
        const float sr = (tid < A) ? raw.s : 0.f;
        const bool clipped = clip_log_std && (sr < min_log_std);   // tf.maximum: gradient iff var >= min
        const float s = clipped ? min_log_std : sr;
        const float sn2 = expf(2.f * s);
        dist[CH_LS + tid] = s;
        dist[CH_ES + tid] = expf(-s);
        dist[CH_SN2 + tid] = sn2;
        dist[CH_LMASK + tid] = clipped ? 0.f : 1.f;
        dist[CH_VLS + tid] = (tid < A && !clipped) ? -raw.v * vs : 0.f;   // tangent along -v (times the direction's scale)
        dist[CH_RDEN + tid] = fast_rcp(2.f * sn2 + 1e-8f);      // one v_rcp_f32 (1 ulp) serves the KL and both of its cotangents
    }
}

// Observation tile in the two operand orientations.  Invalid rows / columns read as 0.
//   xT[t] = X[row i16][col 4 t + kk]           (B operand of layer 1; k-step t)
//   xN[ob][t] = X[row 4 kk + t][col 16 ob + i16]   (A operand of the hidden_0 gradient; k-step t <-> sample 4 kk + t)
template <int KS>
PROMP_DEV void chain_load_xT(float (&xT)[KS], const float* obs, long long row0, int nvalid, int O, int i16, int kk) {
    const bool rv = i16 < nvalid;
    const float* src = obs + (row0 + (rv ? i16 : 0)) * O;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
        const int o = 4 * t + kk;
        const bool ok = rv && o < O;
        // always a valid address, and a multiplicative mask: a select on the loaded value would come back as an exec-masked
        // branch around the load and the loads of a tile would no longer be in flight together
        xT[t] = src[ok ? o : 0] * (ok ? 1.f : 0.f);
    }
}
// this lane's eight observation entries of its sample: obs[row i16][8 kk .. 8 kk + 7] times xs (the FP16 split's observation
// scale), zeros outside the tile / the observation (k_pass: pass_load_x)
PROMP_DEV void chain_load_x8(float (&xr)[8], const float* obs, long long row0, int nvalid, int O, int i16, int kk, float xs) {
    const bool rv = i16 < nvalid;
    const float* src = obs + (row0 + (rv ? i16 : 0)) * O;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int o = 8 * kk + e;
        const bool ok = rv && o < O;
        xr[e] = src[ok ? o : 0] * (ok ? xs : 0.f);
    }
}
template <int NOB>
PROMP_DEV void chain_load_xN(float (&xN)[NOB][4], const float* obs, long long row0, int nvalid, int O, int i16, int kk) {
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = 4 * kk + t, o = 16 * ob + i16;
            const bool ok = r < nvalid && o < O;
            xN[ob][t] = obs[ok ? (row0 + r) * O + o : row0 * O] * (ok ? 1.f : 0.f);
        }
}

PROMP_DEV f32x4 tanh4(f32x4 z) {
    f32x4 h;
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = fast_tanh(z[r]);
    return h;
}

// Cross-wave, fixed-order sum of the waves' gradient tiles -> one partial row in global memory.
// Every wave stores its tiles to its own LDS slab of [NP + 2] floats (plain stores, every entry written exactly once);
// then all threads add the slabs in wave order.  (NW x [NP + 2] floats alias the parameter / transpose regions.)
// W32: the hidden_1 kernel gradient arrives in the 32 x 32 result layout (aw2w) and its bias gradient in the chain layout
// (gb2v: units 16 c + 4 kk + r, already summed over the 16 sample lanes) instead of aw2 / gb2.
// us: the exact power of two that undoes this wave's scales (FP16 split) on the way into the slab; bad: this wave's overflow vote --
// the function returns whether any wave of the workgroup voted.
template <int NC1, int NC2, int NOB, int NW, bool W32 = false>
PROMP_DEV bool chain_reduce_to_partial(float* S, float* P, const f32x4 (&aw2)[NC1][NC2], const f32x16 (&aw2w)[NC1 / 2][NC2 / 2],
                                       const f32x4 (&aw1)[NOB][NC1], const f32x16 (&aw1w)[NC1 / 2], const f32x4 (&gb1v)[NC1],
                                       const f32x4 (&aw3)[NC2], const float (&gb1)[NC1],
                                       const float (&gb2)[NC2], const f32x4 (&gb2v)[NC2], float gs0,
                                       float gs1, float gb30, float gb31, float loss, float klsum, int O, int A, int tid, float us = 1.f,
                                       int bad = 0, float us1 = 1.f) {
    constexpr int H1 = 16 * NC1, H2 = 16 * NC2, NT = 64 * NW;
    // (an opaque copy of the thread index, as in chain_stage_nets: lane-constant indices that live from the kernel's first lines to
    //  this point are spilled across the tile loop, and each reload here is a round trip to scratch memory in the kernel's tail)
    tid += opaque_zero();
    const int lane = tid & 63, w = tid >> 6, i16 = lane & 15, kk = lane >> 4;
    const int ob1 = O * H1, oW2 = ob1 + H1, ob2 = oW2 + H1 * H2, oW3 = ob2 + H2, ob3 = oW3 + H2 * A, oS = ob3 + A, NP = oS + A;
    const int SL = (NP + 2 + 3) & ~3;
    if (lane == 0) ((int*)S)[w - 4] = bad;      // the four words in front of the slabs
    lds_barrier();                // every wave is done with the parameter / transpose regions
    {
        float* mine = S + w * SL;
        if (W32) {
            const int j32 = lane & 31, kh = lane >> 5;
#pragma unroll
            for (int bi = 0; bi < NC1 / 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < NC2 / 2; ++bj)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        mine[oW2 + (32 * bi + (r & 3) + 8 * (r >> 2) + 4 * kh) * H2 + 32 * bj + j32] = aw2w[bi][bj][r] * us;
        } else {
#pragma unroll
            for (int i = 0; i < NC1; ++i)
#pragma unroll
                for (int j = 0; j < NC2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[oW2 + (16 * i + 4 * kk + r) * H2 + 16 * j + i16] = aw2[i][j][r] * us;
        }
        if (W32) {                // (round 6) the hidden_0 kernel gradient in the 32 x 32 result layout: rows = observation slots
            const int j32 = lane & 31, kh = lane >> 5;
#pragma unroll
            for (int bj = 0; bj < NC1 / 2; ++bj)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (row < O) mine[row * H1 + 32 * bj + j32] = aw1w[bj][r] * us1;
                }
        } else {
#pragma unroll
            for (int i = 0; i < NOB; ++i)
#pragma unroll
                for (int j = 0; j < NC1; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * i + 4 * kk + r;
                        if (row < O) mine[row * H1 + 16 * j + i16] = aw1[i][j][r] * us;
                    }
        }
#pragma unroll
        for (int j = 0; j < NC2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (i16 < A) mine[oW3 + (16 * j + 4 * kk + r) * A + i16] = aw3[j][r] * us;
        if (kk == 0) {
            if (!W32) {
#pragma unroll
                for (int j = 0; j < NC1; ++j) mine[ob1 + 16 * j + i16] = gb1[j] * us;
            }
            if (!W32) {
#pragma unroll
                for (int j = 0; j < NC2; ++j) mine[ob2 + 16 * j + i16] = gb2[j] * us;
            }
        }
        if (W32 && i16 == 0) {
#pragma unroll
            for (int j = 0; j < NC2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[ob2 + 16 * j + 4 * kk + r] = gb2v[j][r] * us;
#pragma unroll
            for (int j = 0; j < NC1; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[ob1 + 16 * j + 4 * kk + r] = gb1v[j][r] * us;
        }
        if (i16 == 0) {           // lane (0, kk) holds the sums of actions 2 kk, 2 kk + 1
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

// Sum of a task's partial rows in slot order + the update that consumes it.
struct TaskRedArgs {
    const float* partials;
    int partial_stride, s0, s1, NP, mode, task;
    const float* step_sizes;
    const float* cur;
    long long cur_task_stride;
    float *next, *lam, *v, *scal;
};
template <int NT>
PROMP_DEV void chain_task_sum(TaskRedArgs r, int tid) {
    const int NP = r.NP, mode = r.mode, task = r.task;
    // CW columns per thread, up to 8 rows per round, and the operands of the update (step sizes, current parameters /
    // multipliers) requested in the same batch: the rows live in other CUs' L2 lines or in memory, so the sum is paced by
    // how many loads are outstanding, not by arithmetic
    constexpr int CW = 12;
    const int jbeg = (mode == RED_SCAL) ? NP : 0;
    for (int j0 = jbeg + tid; j0 < NP + 2; j0 += CW * NT) {
        float g[CW], al[CW], old[CW];
#pragma unroll
        for (int u = 0; u < CW; ++u) {
            const int j = j0 + u * NT, jc = j < NP ? j : NP - 1;
            g[u] = 0.f;
            al[u] = r.step_sizes[jc];
            old[u] = (mode == RED_STEP) ? r.cur[(long long)task * r.cur_task_stride + jc]
                                        : (mode == RED_HVP) ? r.lam[(long long)task * NP + jc] : 0.f;
        }
        for (int sb = r.s0; sb < r.s1; sb += 8) {
            float x[8][CW];
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int u = 0; u < CW; ++u) {
                    const int j = j0 + u * NT;
                    const bool ok = sb + q < r.s1 && j < NP + 2;
                    // clamped to a row / column of this task (all written: every workgroup has arrived) and masked by a
                    // multiplication, so that the loads stay unconditional and in flight together
                    const int sq = sb + q < r.s1 ? sb + q : r.s1 - 1, jc = j < NP + 2 ? j : NP + 1;
                    x[q][u] = r.partials[sq * r.partial_stride + jc] * (ok ? 1.f : 0.f);
                }
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int u = 0; u < CW; ++u) g[u] += x[q][u];       // slot order within a column: fixed
        }
#pragma unroll
        for (int u = 0; u < CW; ++u) {
            const int j = j0 + u * NT;
            if (j >= NP + 2) continue;
            if (j >= NP) {
                r.scal[task * 2 + (j - NP)] = g[u];
                continue;
            }
            const long long tj = (long long)task * NP + j;
            if (mode == RED_STEP) {
                r.next[tj] = old[u] - al[u] * g[u];
                continue;
            }
            const float lam = (mode == RED_HVP) ? g[u] + old[u] : g[u];
            r.lam[tj] = lam;
            if (mode == RED_PLAIN) continue;
            r.v[tj] = al[u] * lam;
        }
    }
}

// The workgroup has written a partial row of `task`.  Whichever workgroup of the task arrives last adds the task's
// rows in slot order (bitwise reproducible whatever the arrival order) and applies the update that consumes the sum.
template <int NT>
PROMP_DEV void chain_task_reduce(const PassArgs& a, int* flag, int task, int NP, int tid) {
    const int s0 = a.task_slot_offsets[task], s1 = a.task_slot_offsets[task + 1];
    __syncthreads();                                   // all of this workgroup's partial stores are issued
    if (tid == 0) {
        fence_release_agent();
        const int old = atomic_add_agent(a.task_counters + task, 1);
        const int last = (old == s1 - s0 - 1) ? 1 : 0;
        if (last) {
            atomic_store_agent(a.task_counters + task, 0);   // ready for the next launch
            fence_acquire_agent();
        }
        *flag = last;
    }
    __syncthreads();
    const int last = *flag;
    __syncthreads();                                   // flag may be rewritten by the next segment
    if (!last) return;
    TaskRedArgs r;
    r.partials = a.partials; r.partial_stride = a.partial_stride; r.s0 = s0; r.s1 = s1; r.NP = NP; r.mode = a.red_mode; r.task = task;
    r.step_sizes = a.step_sizes; r.cur = a.cur; r.cur_task_stride = a.cur_task_stride;
    r.next = a.next; r.lam = a.lam; r.v = a.v; r.scal = a.scal;
    chain_task_sum<NT>(r, tid);
}

// FP16 split of k_chain_hvp (see pass_cotangent_scale in promp_kernels_pass.h for the reasoning): the direction's largest entry
// (hidden_0 block weighed by the observations' scale) goes to [8, 16) -- tangent activations, sums over the observations, then sit
// around 1 .. 1000 --, the larger of the first tile's largest primal cotangent and 1/32 of its largest tangent cotangent to [4, 8).
// Measured on configs 3 and 4 with the direction's target at 2, 8 and 32 (profiles/r06_split_targets.txt): no segment walked again
// in any of them, the meta-gradient's error at 128-wide layers 4.7 / 3.5 / 3.2 e-6.  A wave that finds an infinity or a NaN in its
// sums -- every overflow of a split ends there -- has the workgroup walk the segment again: the largest cotangent -- known by
// then -- at 2^CHAIN_CT_REDO, and on a third walk the direction 2^CHAIN_V_RETRY lower as well (tangent activations that left the
// format: observations x direction beyond 2^16).  One such segment doubles the launch's duration (the other workgroups wait).
#ifndef PROMP_CT_ATTEMPTS
#define PROMP_CT_ATTEMPTS 3
#endif
#ifndef PROMP_CHAIN_V_TARGET
#define PROMP_CHAIN_V_TARGET 3
#endif
#ifndef PROMP_CHAIN_CT_TARGET
#define PROMP_CHAIN_CT_TARGET 2
#endif
PROMP_CX int CHAIN_V_TARGET = PROMP_CHAIN_V_TARGET, CHAIN_V_RETRY = 10, CHAIN_CT_TARGET = PROMP_CHAIN_CT_TARGET,
             CHAIN_CT_REDO = 8, CHAIN_CT_RETRY = 10, CHAIN_Q_OVER_D = 5, CHAIN_ATTEMPTS = PROMP_CT_ATTEMPTS;

// Two-network product of the R-operator pass on the BF16 pipe, operands streamed:
//     acc[c] += W[.][c] xa + V[.][c] xb          (PRIMAL: accp[c] += W[.][c] xb as well)
// W / V: the BF16 planes [term][c < NCO][chunk P < NPK][lane] of theta's and the direction's kernel, xa / xb: the planes of the two
// activations (a lane's eight k-slots of chunk P).  Six of the nine term products, walked by weight term (2,0) (1,1) (1,0) (0,2)
// (0,1) (0,0) -- roughly smallest first -- so that a weight fragment is read ONCE and feeds one to three consecutive groups of
// instructions (FP16 split, round 6: two terms, (1,0) (0,1) (0,0)), and only two terms' fragments are in registers at a time: the fragments of group g + 1 are requested before the
// products of group g are issued (left alone the compiler reads a fragment right in front of its first product, and the single wave
// of a SIMD sits out one LDS latency per fragment: ~45 cycles x 35 waits per tile in each of the two K = 64 products, measured).
template <int NCO, int NPK, bool PRIMAL>
PROMP_DEV void chain_gemm2_bf16(f32x4 (&acc)[NCO], f32x4 (&accp)[NCO], const u32x4* Wp, const u32x4* Vp, const u32x4 (&xa)[NPK][PROMP_NT],
                                const u32x4 (&xb)[NPK][PROMP_NT]) {
    constexpr int NT = PROMP_NT, NG = NT * NPK, NM = (PRIMAL ? 3 : 2) * NCO;
    u32x4 fw[2][NCO], fv[2][NCO];
#pragma unroll
    for (int c = 0; c < NCO; ++c) {
        fw[0][c] = Wp[(((NT - 1) * NCO + c) * NPK + 0) * 64];
        fv[0][c] = Vp[(((NT - 1) * NCO + c) * NPK + 0) * 64];
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int P = g / NT, ta = NT - 1 - g % NT, cur = g & 1, nxt = cur ^ 1;
        if (g + 1 < NG) {
            const int Pn = (g + 1) / NT, tan = NT - 1 - (g + 1) % NT;
#pragma unroll
            for (int c = 0; c < NCO; ++c) {
                fw[nxt][c] = Wp[((tan * NCO + c) * NPK + Pn) * 64];
                fv[nxt][c] = Vp[((tan * NCO + c) * NPK + Pn) * 64];
            }
        }
#pragma unroll
        for (int tb = NT - 1 - ta; tb >= 0; --tb) {
            if (PRIMAL) {
#pragma unroll
                for (int c = 0; c < NCO; ++c) accp[c] = mfma16_sw<NT>(fw[cur][c], xb[P][tb], accp[c]);
            }
#pragma unroll
            for (int c = 0; c < NCO; ++c) acc[c] = mfma16_sw<NT>(fw[cur][c], xa[P][tb], acc[c]);
#pragma unroll
            for (int c = 0; c < NCO; ++c) acc[c] = mfma16_sw<NT>(fv[cur][c], xb[P][tb], acc[c]);
        }
        if (g + 1 < NG) PROMP_SCHED_DSREAD(2 * NCO);        // the requests first, then the products
        if (NT - ta == 1) PROMP_SCHED_MFMA(NM);
        else if (NT - ta == 2) PROMP_SCHED_MFMA(2 * NM);
        else PROMP_SCHED_MFMA(3 * NM);
        sched_fence();
    }
}

// ---------------------------------------------------------------------------------------------
// k_chain_hvp:  out = -(d^2 L/d theta^2) v + kl_weight * grad KL   (R-operator, see oracle/promp.py:hvp)
//
// The direction is staged NEGATED, so every tangent below is the derivative along -v ("R'"): the reverse-pass
// quantities q = R'{.} + kl_weight * dKL{.} then need no operand negation anywhere (MFMA has no negate modifier for
// f32 operands).  theta and -v are both staged in fragment order (2 x 32.5 KB at 64/64).
// ---------------------------------------------------------------------------------------------
//
// CACHED: the primal activations and means of every tile come from the step's primal cache (written by the gradient pass at
// the same parameters, see chain_cache_row) instead of being recomputed: the primal products of the three layers and the
// tanh evaluations drop out; the next tile's block is requested half a tile ahead.
template <int NC1, int NC2, int KS, int NW, bool CACHED = false>
__global__ void __launch_bounds__(64 * NW, NW / 4) k_chain_hvp(PassArgs a) {
    constexpr int NT = 64 * NW, H1 = 16 * NC1, H2 = 16 * NC2, TS = PROMP_CH_TS, DS = PROMP_CH_DS, NOB = KS > 4 ? 2 : 1;
    constexpr int HCR = chain_cache_row(H1, H2);
    constexpr ChainLds L = chain_layout(NC1, NC2, NW, true, 0, CACHED);      // CACHED: the layout with backward planes
    PROMP_SMEM_DECL;
    float* sm = (float*)PROMP_SMEM_PTR;
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
    const int i16 = lane & 15, kk = lane >> 4;
    const int O = a.O, A = a.A;
    const int ob1 = O * H1, oW2 = ob1 + H1, ob2 = oW2 + H1 * H2, oW3 = ob2 + H2, ob3 = oW3 + H2 * A, oS = ob3 + A, NP = oS + A;
    float* net = sm + 4;
    float* vnet = net + L.net_stride;
    constexpr int VO = L.net_stride;                  // the -v fragments sit at the same offsets, one block further
    float* wreg = sm + L.wave0 + w * L.wave_stride;
    float *TB0 = wreg + L.tb0, *TB1 = wreg + L.tb1, *DB0 = wreg + L.db0, *DB1 = wreg + L.db1;
    const float* W1l = net + L.w1 + lane * 4;
    const float* W2l = net + L.w2 + kk * PROMP_CH_ROW + i16 * 4;
    const float* W2b = net + L.w2 + (i16 >> 2) * PROMP_CH_ROW + (i16 & 3) + 16 * kk;
    // output kernel fragments: by action slot (one per lane), or -- CACHED -- by action: lanes of a padding slot read a valid
    // fragment and multiply it by zero
    constexpr int W3C = 256;
    const float* W3l = net + L.w3 + lane * 4;        // (!CACHED: float32 fragments by action slot)
    const float* W3b = net + L.w3b + lane * 2;
    // CACHED (round 6): the output kernels and the direction's hidden_0 kernel as the split's planes in k_pass's fragment order
    const u32x4* W3F = (const u32x4*)(net + L.w3) + lane;
    const u32x4* V3F = (const u32x4*)(vnet + L.w3) + lane;
    const u32x4* V1F = (const u32x4*)(vnet + L.w1) + lane;
    const float *B1l = net + L.b1 + 4 * kk, *B2l = net + L.b2 + 4 * kk, *B3l = net + L.b3 + 2 * kk;
    float* TB0w = TB0 + i16 * TS + 4 * kk;
    float* TB1w = TB1 + i16 * TS + 4 * kk;
    const float* TB0r = TB0 + kk * 4 * TS + i16;
    const float* TB1r = TB1 + kk * 4 * TS + i16;
    float* DB0w = DB0 + i16 * DS + 2 * kk;
    float* DB1w = DB1 + i16 * DS + 2 * kk;
    const float* DB0r = DB0 + kk * 4 * DS + i16;
    const float* DB1r = DB1 + kk * 4 * DS + i16;
    const int a0 = 2 * kk, a1 = 2 * kk + 1;
    const bool own0 = a0 < A, own1 = a1 < A;
    const int q0 = own0 ? a0 : 0, q1 = own1 ? a1 : 0;
    const float klw = a.kl_weight;
    // CACHED: the hidden_1 kernel gradient runs on v_mfma_f32_32x32x16_bf16 (round 4): its operands' BF16 planes go through
    // swizzled tiles over TB0 / TB1 and come back through the transpose read (pass_slot; word addresses of this lane)
    constexpr int NB1 = NC1 / 2, NB2 = NC2 / 2, TPL = PROMP_CH_TPL, APL = PROMP_CH_APL;
    float *TBP = wreg + L.tp;
    // ... and (round 6) every other product of the cached instance too: k_pass's tiles and lane addresses
    constexpr int XPL = PROMP_PASS_XPLANE, DPL = PROMP_PASS_DPLANE, NP1 = NC1 / 2, NP2 = NC2 / 2;
    float *TQ = wreg + L.tq, *XT = wreg + L.xt, *DM0 = wreg + L.dm0, *DM1 = wreg + L.dm1;
    const PassTileAddr T = pass_tile_addr(lane);
    const int twr = 2 * pass_slot(i16, kk);
    const int trd0 = 2 * pass_slot(8 * (lane >> 5) + (i16 >> 2), 4 * (kk & 1) + (i16 & 3));
    const int trd1 = 2 * pass_slot(8 * (lane >> 5) + 4 + (i16 >> 2), 4 * (kk & 1) + (i16 & 3));

    const int wgi = PROMP_PASS_XCD ? xcd_item(blockIdx.x, gridDim.x) : (int)blockIdx.x;     // (see PROMP_PASS_XCD)
    const int sg0 = a.wg_seg_offsets[wgi], sg1 = a.wg_seg_offsets[wgi + 1];
    CH_WGSTAMP(0);
    int attempt = 0;              // FP16 split: how often the current segment has overflowed (see the end of the tile walk)
    float redo_amax = 0.f;        // ... and the largest cotangent this wave met on the way
    for (int sg = sg0; sg < sg1; ++sg) {
        const ChainSeg seg = a.segs[sg];
        const int task = seg.task;
        const int trow0 = a.task_row_offsets[task], tnrows = a.task_row_offsets[task + 1] - trow0;
        const float invN = 1.0f / (float)tnrows;
        const float* th = a.theta + (long long)task * a.theta_task_stride;
        const float* v = a.vdir + (long long)task * NP;
        // The segment's requests, in the order their answers are needed: the distribution's raw parameters and the networks (from
        // L2), then -- inside the staging, behind its last load -- the first tile's observations and (CACHED) its cache block (from
        // memory); the workgroup joins (the previous segment is done with LDS) only after everything is on its way.
        const ChainDistRaw draw = chain_dist_load(th, v, oS, A, tid);
        const int tend = seg.tile0 + seg.ntiles;
        float xT[KS];
        float xr[8];              // CACHED: the tile's observations as k_pass reads them (eight consecutive slots of the lane's sample)
        const int sx = obs_shift(a.obs_absmax, task);
        const float xs = pow2f(-sx), w1w = pow2f(sx);
        // CACHED: this lane's share of a tile's cache block (sample i16, units 16 c + 4 kk + r; actions 2 kk, 2 kk + 1)
        f32x4 ch1[NC1], ch2[NC2];
        f32x2 cmu;
        const float* hcl = CACHED ? a.hcache + ((long long)trow0 + 16 * task) * HCR + i16 * 16 + 4 * kk : nullptr;
        const float* hcm = CACHED ? a.hcache + ((long long)trow0 + 16 * task) * HCR + 256 * (NC1 + NC2) + i16 * 8 + 2 * kk : nullptr;
        CH_STAMP(0);
        // FP16 split: where the largest direction entry / the first tile's largest mean cotangent go (chain_stage_nets; below)
        // (the second walk keeps the direction's scale -- heavy-tailed cotangents are the usual reason, and the known maximum settles
        //  them --, the third lowers it too)
        const int vt = CHAIN_V_TARGET - (attempt >= 2 ? CHAIN_V_RETRY : 0), ct = attempt ? CHAIN_CT_REDO - (attempt - 1) * CHAIN_CT_RETRY : CHAIN_CT_TARGET;
        // (L is the layout for a parameter count of 0: the slabs of the real one may end behind L.vmx; the host sized LDS for both)
        const int slabs_end = 4 + NW * ((NP + 2 + 3) & ~3), vmx_off = slabs_end > L.vmx ? slabs_end : L.vmx;
        const float vs = chain_stage_nets<NC1, NC2, NW, CACHED>(sm, sm + vmx_off, th, v, O, A, tid, vt, w1w, [&]() {
            {
                const int t = seg.tile0 + w;
                const int nv = (t < tend) ? (tnrows - 16 * t < 16 ? tnrows - 16 * t : 16) : 0;
                if (CACHED) chain_load_x8(xr, a.obs, (long long)trow0 + (t < tend ? 16 * t : 0), nv, O, i16, kk, xs);
                else chain_load_xT<KS>(xT, a.obs, (long long)trow0 + (t < tend ? 16 * t : 0), nv, O, i16, kk);
            }
            if (CACHED) {
                const int t = seg.tile0 + w;
                const long long o = (long long)(t < tend ? 16 * t : 16 * seg.tile0) * HCR;     // always a block of this segment
#pragma unroll
                for (int c = 0; c < NC1; ++c) ch1[c] = *(const f32x4*)(hcl + o + 256 * c);
#pragma unroll
                for (int c = 0; c < NC2; ++c) ch2[c] = *(const f32x4*)(hcl + o + 256 * (NC1 + c));
                cmu = *(const f32x2*)(hcm + o);
            }
        });
        CH_STAMP(5);
        chain_stage_dist(net + L.dist, draw, A, a.clip_log_std, a.min_log_std, tid, vs);
        CH_STAMP(6);
        // action slots >= 8 of the cotangent tiles must read as zero (slots < 8 and the transpose tiles are rewritten by
        // every tile before they are read); the end-of-segment slabs alias them, so once per segment
        if (CACHED) {
            for (int e = lane; e < 2 * PROMP_NT * DPL; e += 64) DM0[e] = 0.f;       // (DM1 follows DM0)
        } else {
            for (int e = lane; e < 2 * 16 * DS; e += 64) DB0[e] = 0.f;
        }
        CH_STAMP(7);
        __syncthreads();
        CH_STAMP(1);
        const float* dist = net + L.dist;
        const float s0 = dist[CH_LS + q0], s1 = dist[CH_LS + q1], e0 = dist[CH_ES + q0], e1 = dist[CH_ES + q1];
        const float sn20 = dist[CH_SN2 + q0], sn21 = dist[CH_SN2 + q1], rden0 = dist[CH_RDEN + q0], rden1 = dist[CH_RDEN + q1];
        const float Rs0 = dist[CH_VLS + q0], Rs1 = dist[CH_VLS + q1];       // R'{s}

        f32x4 aw2[NC1][NC2], aw1[NOB][NC1], aw3[NC2];
#pragma unroll
        for (int i = 0; i < NC1; ++i)
#pragma unroll
            for (int j = 0; j < NC2; ++j) aw2[i][j] = zero4();
#pragma unroll
        for (int i = 0; i < NOB; ++i)
#pragma unroll
            for (int j = 0; j < NC1; ++j) aw1[i][j] = zero4();
#pragma unroll
        for (int j = 0; j < NC2; ++j) aw3[j] = zero4();
        float ob1acc[NC1], ob2acc[NC2];
#pragma unroll
        for (int j = 0; j < NC1; ++j) ob1acc[j] = 0.f;
#pragma unroll
        for (int j = 0; j < NC2; ++j) ob2acc[j] = 0.f;
        // CACHED: the hidden_1 and hidden_0 kernel gradients in the 32 x 32 result layout, their bias gradients in the chain layout
        f32x16 aw2w[NB1][NB2], aw1w[NB1];
        f32x4 gb2v[NC2], gb1v[NC1];
#pragma unroll
        for (int i = 0; i < NB1; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) aw1w[i][r] = 0.f;
#pragma unroll
        for (int j = 0; j < NC1; ++j) gb1v[j] = zero4();
#pragma unroll
        for (int i = 0; i < NB1; ++i)
#pragma unroll
            for (int j = 0; j < NB2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) aw2w[i][j][r] = 0.f;
#pragma unroll
        for (int j = 0; j < NC2; ++j) gb2v[j] = zero4();
        float klsum = 0.f, outs0 = 0.f, outs1 = 0.f, outb30 = 0.f, outb31 = 0.f;
        // FP16 split: this wave's cotangent scale in the segment (set by its first tile with a cotangent, as in k_pass), the tangent
        // cotangents' (they carry the direction's scale too) and the weight of the KL cotangents among them
        float cs = 1.f, qs = vs, ivs = 1.f / vs, amax = 0.f;
        int prov = 1;
        if (PROMP_NT == 2 && attempt > 0 && redo_amax > 0.f && redo_amax < 3.0e38f) {      // the segment again: the largest cotangent is known
            int k = scale_exp(redo_amax, CHAIN_CT_REDO - (attempt - 1) * CHAIN_CT_RETRY);
            k = k < -100 ? -100 : k > 100 ? 100 : k;
            cs = pow2f(k);
            qs = cs * vs;
            prov = 0;
        }

        int tix = 0;
        for (int t = seg.tile0 + w; t < tend; t += NW, ++tix) {
            CH_TSTAMP(0);
            const int nrows = (tnrows - 16 * t) < 16 ? (tnrows - 16 * t) : 16;
            const long long base = (long long)trow0 + 16 * t;
            const bool rvalid = i16 < nrows;
            const long long n = base + (rvalid ? i16 : 0);
            const float* olsp = a.old_log_std + (a.ls_per_row ? n * A : (long long)task * A);
            const float advn = rvalid ? a.adv[n] : 0.f;
            const float ac0 = a.act[n * A + q0], ac1 = a.act[n * A + q1];
            const float mo0 = a.old_mean[n * A + q0], mo1 = a.old_mean[n * A + q1];
            const float so0 = olsp[q0], so1 = olsp[q1];

            f32x4 cad1[NC1];      // CACHED: W2 dZ2^T of this tile, needed at the very end of the tile
            if (CACHED) {
                const long long o = (long long)16 * t * HCR + 256 * (NC1 + NC2) + 128;
#pragma unroll
                for (int c = 0; c < NC1; ++c) cad1[c] = *(const f32x4*)(hcl + o + 256 * c);
            }
            // ---- layer 1 and its tangent:  R'z1 = X (-vW1) + (-vb1)
            f32x4 h1[NC1], rh1[NC1];
            {
#pragma unroll
                for (int c = 0; c < NC1; ++c) {
                    h1[c] = CACHED ? ch1[c] : lds4(B1l + 16 * c);
                    rh1[c] = lds4(B1l + VO + 16 * c);
                }
                if (CACHED) {
                    // (round 6) on the split pipe, k_pass's layer 1: the observations' planes (kept in their tile for the hidden_0
                    // kernel gradient at the end of the tile) against the direction's hidden_0 planes, K = 32 observation slots
                    u32x4 xB[PROMP_NT], v1f[PROMP_NT][NC1];
                    {
                        f32x4 lo, hi;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            lo[e] = xr[e];
                            hi[e] = xr[4 + e];
                        }
                        pass_split8(lo, hi, xB);
                    }
#pragma unroll
                    for (int ta = 0; ta < PROMP_NT; ++ta)
#pragma unroll
                        for (int c = 0; c < NC1; ++c) v1f[ta][c] = V1F[(ta * NC1 + c) * 64];
                    wave_fence();         // the previous tile's reads of the observation tile precede these writes
#pragma unroll
                    for (int tt = 0; tt < PROMP_NT; ++tt) {
                        sts_w2(XT + tt * XPL + T.xw0, xB[tt][0], xB[tt][1]);
                        sts_w2(XT + tt * XPL + T.xw1, xB[tt][2], xB[tt][3]);
                    }
#pragma unroll
                    for (int ta = PROMP_NT - 1; ta >= 0; --ta)
#pragma unroll
                        for (int tb = PROMP_NT - 1 - ta; tb >= 0; --tb)
#pragma unroll
                            for (int c = 0; c < NC1; ++c) rh1[c] = mfma16_sw<PROMP_NT>(v1f[ta][c], xB[tb], rh1[c]);
                } else {
#pragma unroll
                for (int t4 = 0; t4 < NOB; ++t4) {
                    f32x4 wf[NC1], vf[NC1];
#pragma unroll
                    for (int c = 0; c < NC1; ++c) {
                        wf[c] = lds4(W1l + (2 * c + t4) * 256);
                        vf[c] = lds4(W1l + VO + (2 * c + t4) * 256);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * t4 + r < KS) {
#pragma unroll
                            for (int c = 0; c < NC1; ++c) h1[c] = mfma16(wf[c][r], xT[4 * t4 + r], h1[c]);
#pragma unroll
                            for (int c = 0; c < NC1; ++c) rh1[c] = mfma16(vf[c][r], xT[4 * t4 + r], rh1[c]);
                        }
                }
                }
#pragma unroll
                for (int c = 0; c < NC1; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float h = CACHED ? h1[c][r] : fast_tanh(h1[c][r]);
                        h1[c][r] = h;
                        rh1[c][r] *= (1.f - h * h);
                    }
            }
            {
                const int tn = t + NW;
                const int nv = (tn < tend) ? (tnrows - 16 * tn < 16 ? tnrows - 16 * tn : 16) : 0;
                if (CACHED) chain_load_x8(xr, a.obs, (long long)trow0 + (tn < tend ? 16 * tn : 0), nv, O, i16, kk, xs);
                else chain_load_xT<KS>(xT, a.obs, (long long)trow0 + (tn < tend ? 16 * tn : 0), nv, O, i16, kk);
            }
            CH_TSTAMP(1);
            // ---- layer 2 and its tangent:  R'z2 = W2^T R'H1 + (-vW2)^T H1 + (-vb2)
            f32x4 h2[NC2], rh2[NC2];
            {
#pragma unroll
                for (int c = 0; c < NC2; ++c) {
                    h2[c] = CACHED ? ch2[c] : lds4(B2l + 16 * c);
                    rh2[c] = lds4(B2l + VO + 16 * c);
                }
                {
                    // R'z2 += W2^T R'H1 + (-vW2)^T H1 on the BF16 pipe: K = 32 per instruction = two 16-unit input blocks; a
                    // lane's eight k-slots are its own registers of the two blocks (units 16 c + 4 kk + r), split three ways
                    // (!CACHED: the primal product z2 = W2^T H1 rides on the same planes and splits)
                    u32x4 rB[NC1 / 2][PROMP_NT], hB[NC1 / 2][PROMP_NT];
#pragma unroll
                    for (int P = 0; P < NC1 / 2; ++P) {
                        pass_split8(rh1[2 * P], rh1[2 * P + 1], rB[P]);
                        pass_split8(h1[2 * P], h1[2 * P + 1], hB[P]);
                    }
                    chain_gemm2_bf16<NC2, NC1 / 2, !CACHED>(rh2, h2, (const u32x4*)(sm + L.planes) + lane,
                                                            (const u32x4*)(sm + L.planes + L.plane_stride) + lane, rB, hB);
                }
#pragma unroll
                for (int c = 0; c < NC2; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float h = CACHED ? h2[c][r] : fast_tanh(h2[c][r]);
                        h2[c][r] = h;
                        rh2[c][r] *= (1.f - h * h);
                    }
            }
            CH_TSTAMP(2);
            // ---- output layer and its tangent:  R'mu = W3^T R'H2 + (-vW3)^T H2 + (-vb3)
            float mu0, mu1, Rmu0, Rmu1;
            if (CACHED) {
                // (round 6) on the split pipe, k_pass's output layer: the planes of R'H2 / H2 feed the two products here (rows of
                // the product: action slots, the lane's registers 0, 1 = actions 2 kk, 2 kk + 1) and, through their tiles and the
                // transpose read, the output-kernel gradient below
                u32x4 rB2[NP2][PROMP_NT], hB2[NP2][PROMP_NT];
#pragma unroll
                for (int P = 0; P < NP2; ++P) {
                    pass_split8(rh2[2 * P], rh2[2 * P + 1], rB2[P]);
                    pass_split8(h2[2 * P], h2[2 * P + 1], hB2[P]);
                }
                wave_fence();         // the previous tile's reads of the plane tiles precede these writes
#pragma unroll
                for (int P = 0; P < NP2; ++P) {
                    pass_store_planes(TBP, TPL, twr + 256 * P, rB2[P]);
                    pass_store_planes(TQ, TPL, twr + 256 * P, hB2[P]);
                }
                f32x4 ra[2] = {zero4(), zero4()};
                const f32x2 vb = lds2(B3l + VO);
                ra[0][0] = vb[0];
                ra[0][1] = vb[1];
#pragma unroll
                for (int P = 0; P < NP2; ++P) {
                    u32x4 w3f[PROMP_NT], v3f[PROMP_NT];
#pragma unroll
                    for (int ta = 0; ta < PROMP_NT; ++ta) {
                        w3f[ta] = W3F[(ta * NP2 + P) * 64];
                        v3f[ta] = V3F[(ta * NP2 + P) * 64];
                    }
#pragma unroll
                    for (int ta = PROMP_NT - 1; ta >= 0; --ta)
#pragma unroll
                        for (int tb = PROMP_NT - 1 - ta; tb >= 0; --tb) {
                            ra[0] = mfma16_sw<PROMP_NT>(w3f[ta], rB2[P][tb], ra[0]);
                            ra[1] = mfma16_sw<PROMP_NT>(v3f[ta], hB2[P][tb], ra[1]);
                        }
                }
                mu0 = cmu[0];
                mu1 = cmu[1];
                Rmu0 = ra[0][0] + ra[1][0];
                Rmu1 = ra[0][1] + ra[1][1];
            } else {
                wave_fence();         // the previous tile's reads of TB0 / TB1 precede these writes
#pragma unroll
                for (int c = 0; c < NC2; ++c) {
                    sts4(TB0w + 16 * c, rh2[c]);
                    sts4(TB1w + 16 * c, h2[c]);
                }
                f32x4 m0 = zero4(), m1 = zero4(), ra0 = zero4(), ra1 = zero4(), rb0 = zero4(), rb1 = zero4();
                const f32x2 bb = lds2(B3l), vb = lds2(B3l + VO);
                m0[0] = bb[0];
                m0[1] = bb[1];
                ra0[0] = vb[0];
                ra0[1] = vb[1];
                f32x4 wf[NC2], vf[NC2];
#pragma unroll
                for (int c = 0; c < NC2; ++c) {
                    wf[c] = lds4(W3l + c * W3C);
                    vf[c] = lds4(W3l + VO + c * W3C);
                }
#pragma unroll
                for (int c = 0; c < NC2; ++c) {
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        m0 = mfma16(wf[c][r], h2[c][r], m0);
                        ra0 = mfma16(wf[c][r], rh2[c][r], ra0);
                        rb0 = mfma16(vf[c][r], h2[c][r], rb0);
                        m1 = mfma16(wf[c][r + 1], h2[c][r + 1], m1);
                        ra1 = mfma16(wf[c][r + 1], rh2[c][r + 1], ra1);
                        rb1 = mfma16(vf[c][r + 1], h2[c][r + 1], rb1);
                    }
                }
                mu0 = m0[0] + m1[0];
                mu1 = m0[1] + m1[1];
                Rmu0 = (ra0[0] + ra1[0]) + (rb0[0] + rb1[0]);
                Rmu1 = (ra0[1] + ra1[1]) + (rb0[1] + rb1[1]);
            }
            CH_TSTAMP(3);
            // ---- loss-level R-operator: lane (i16, kk) = sample i16, actions 2 kk and 2 kk + 1
            float d0, d1, qm0, qm1;
            {
                const float z0 = (ac0 - mu0) * e0, z1 = (ac1 - mu1) * e1;
                const float zo0 = (ac0 - mo0) * fast_exp(-so0), zo1 = (ac1 - mo1) * fast_exp(-so1);
                const float num0 = (mo0 - mu0) * (mo0 - mu0) + fast_exp(2.f * so0) - sn20;
                const float num1 = (mo1 - mu1) * (mo1 - mu1) + fast_exp(2.f * so1) - sn21;
                const float den0 = 2.f * sn20 + 1e-8f, den1 = 2.f * sn21 + 1e-8f;
                float dlp = (own0 ? (so0 - s0) - 0.5f * (z0 * z0 - zo0 * zo0) : 0.f) + (own1 ? (so1 - s1) - 0.5f * (z1 * z1 - zo1 * zo1) : 0.f);
                float Rlp = (own0 ? z0 * e0 * Rmu0 + (z0 * z0 - 1.f) * Rs0 : 0.f) + (own1 ? z1 * e1 * Rmu1 + (z1 * z1 - 1.f) * Rs1 : 0.f);
                float kl = (own0 ? num0 * rden0 + s0 - so0 : 0.f) + (own1 ? num1 * rden1 + s1 - so1 : 0.f);
                dlp = fold_groups16(dlp);      // sums over the row's actions (the four lane groups): lane swaps, no LDS round trip
                Rlp = fold_groups16(Rlp);
                kl = fold_groups16(kl);
                if (a.row_tan != nullptr && rvalid && kk == 0) a.row_tan[n] = Rlp * ivs;
                float c = 0.f, km = 0.f;
                if (rvalid) {
                    km = 1.f;
                    c = (a.loss_kind == LOSS_RATIO) ? -advn * expf(dlp) * invN : -advn * invN;
                }
                const float dklm0 = -2.f * (mo0 - mu0) * rden0 * invN, dklm1 = -2.f * (mo1 - mu1) * rden1 * invN;
                const float dkls0 = ((-2.f * sn20 * den0 - 4.f * num0 * sn20) * (rden0 * rden0) + 1.f) * invN;
                const float dkls1 = ((-2.f * sn21 * den1 - 4.f * num1 * sn21) * (rden1 * rden1) + 1.f) * invN;
                const bool is_kl = a.loss_kind == LOSS_KL;
                const float Rc = (a.loss_kind == LOSS_RATIO) ? c * Rlp : 0.f;
                const float klv = klw * vs;          // (the KL cotangents carry no factor of the direction: they take its scale here)
                // the cotangents of the mean -- primal d (unscaled), tangent q (at the direction's scale vs) -- and the tangent
                // cotangents of log_std, os
                float os0, os1;
                if (is_kl) {
                    // The objective is the mean KL itself (the TRPO constraint): primal cotangent dKL/dmu, tangent cotangents
                    // R'{dKL/dmu}, R'{dKL/ds}.  With D = mu_old - mu, den = 2 e^{2s} + 1e-8, num = D^2 + e^{2 s_old} - e^{2s}:
                    //   dKL/dmu = -2 D / den                     R'{.} = 2 R'mu / den + 8 D e^{2s} R's / den^2
                    //   dKL/ds  = 1 - 2 P / den^2, P = e^{2s} (den + 2 num)
                    //   R'{P}   = 2 e^{2s} R's (den + 2 num) - 4 e^{2s} D R'mu ;  R'{dKL/ds} = -2 R'{P} / den^2 + 16 P e^{2s} R's / den^3
                    {
                        const float D = mo0 - mu0, P = sn20 * (den0 + 2.f * num0);
                        const float RP = 2.f * sn20 * Rs0 * (den0 + 2.f * num0) - 4.f * sn20 * D * Rmu0;
                        const float Rdm = 2.f * Rmu0 * rden0 + 8.f * D * sn20 * Rs0 * (rden0 * rden0);
                        const float Rds = (-2.f * RP + 16.f * P * sn20 * Rs0 * rden0) * (rden0 * rden0);
                        d0 = own0 ? km * dklm0 : 0.f;
                        qm0 = own0 ? km * Rdm * invN : 0.f;
                        os0 = own0 ? km * Rds * invN : 0.f;
                    }
                    {
                        const float D = mo1 - mu1, P = sn21 * (den1 + 2.f * num1);
                        const float RP = 2.f * sn21 * Rs1 * (den1 + 2.f * num1) - 4.f * sn21 * D * Rmu1;
                        const float Rdm = 2.f * Rmu1 * rden1 + 8.f * D * sn21 * Rs1 * (rden1 * rden1);
                        const float Rds = (-2.f * RP + 16.f * P * sn21 * Rs1 * rden1) * (rden1 * rden1);
                        d1 = own1 ? km * dklm1 : 0.f;
                        qm1 = own1 ? km * Rdm * invN : 0.f;
                        os1 = own1 ? km * Rds * invN : 0.f;
                    }
                } else {
                    {
                        const float Rz = -Rmu0 * e0 - z0 * Rs0;
                        const float Rd = Rc * z0 * e0 + c * (Rz * e0 - z0 * e0 * Rs0);
                        const float Rds = Rc * (z0 * z0 - 1.f) + 2.f * c * z0 * Rz;
                        d0 = own0 ? c * z0 * e0 : 0.f;
                        qm0 = own0 ? km * (Rd + klv * dklm0) : 0.f;
                        os0 = own0 ? km * (Rds + klv * dkls0) : 0.f;
                    }
                    {
                        const float Rz = -Rmu1 * e1 - z1 * Rs1;
                        const float Rd = Rc * z1 * e1 + c * (Rz * e1 - z1 * e1 * Rs1);
                        const float Rds = Rc * (z1 * z1 - 1.f) + 2.f * c * z1 * Rz;
                        d1 = own1 ? c * z1 * e1 : 0.f;
                        qm1 = own1 ? km * (Rd + klv * dklm1) : 0.f;
                        os1 = own1 ? km * (Rds + klv * dkls1) : 0.f;
                    }
                }
                if (PROMP_NT == 2) {
                    // FP16 has a range (promp_kernels_pass.h: pass_cotangent_scale): the wave's first tile with a cotangent sets the
                    // power of two cs every cotangent of the segment carries (the tangent cotangents: cs vs).  What has to fit is the
                    // larger of the primal cotangent and 2^-CHAIN_Q_OVER_D of the tangent one -- at TRPO's operating point the primal
                    // cotangent of the KL is rounding noise and the tangent one is everything.  The tile pays two instructions for the
                    // wave's running maximum; what it means is settled at the end of the segment.
                    const float am = fmaxf(fmaxf(fabsf(d0), fabsf(d1)), (1.f / (float)(1 << CHAIN_Q_OVER_D)) * fmaxf(fabsf(qm0), fabsf(qm1)));
                    amax = fmaxf(amax, am);
                    if (wave_uniform(prov)) {
                        const float mx = wave_absmax_f32(am);
                        const bool okm = mx > 0.f && mx < 3.0e38f;
                        int k = scale_exp(okm ? mx : invN, okm ? ct : -4);
                        k = k < -100 ? -100 : k > 100 ? 100 : k;
                        cs = pow2f(k);
                        qs = cs * vs;
                        prov = okm ? 0 : 1;
                    }
                    d0 *= cs; d1 *= cs; qm0 *= cs; qm1 *= cs; os0 *= cs; os1 *= cs;
                }
                if (rvalid && kk == 0) klsum += kl * invN;
                outs0 += os0;
                outs1 += os1;
                outb30 += qm0;
                outb31 += qm1;
            }
            CH_TSTAMP(4);
            // ---- out_W3 += R'H2^T dmu + H2^T qmu
            if (CACHED) {
                // (round 6) on the split pipe, k_pass's output-kernel gradient: the cotangents of the mean through two small plane
                // tiles, R'H2 / H2 back from their tiles through the transpose read (16-unit blocks, K = 16 samples zero-padded to 32)
                unsigned dw[PROMP_NT], qw[PROMP_NT];
                split_pair<PROMP_NT>(d0, d1, dw);
                split_pair<PROMP_NT>(qm0, qm1, qw);
#pragma unroll
                for (int tt = 0; tt < PROMP_NT; ++tt) {
                    DM0[tt * DPL + T.dmw] = __builtin_bit_cast(float, dw[tt]);
                    DM1[tt * DPL + T.dmw] = __builtin_bit_cast(float, qw[tt]);
                }
                wave_fence();
                u32x4 bD[PROMP_NT], bQ[PROMP_NT];
                pass_read_tr(bD, DM0, DPL, T.dr0, T.dr1);
                pass_read_tr(bQ, DM1, DPL, T.dr0, T.dr1);
#pragma unroll
                for (int c = 0; c < NC2; ++c) {
                    u32x4 aR[PROMP_NT], aH[PROMP_NT];
                    const int off = 2 * (128 * (c >> 1) + 16 * (c & 1));
                    pass_read_tr(aR, TBP, TPL, T.rd16_0 + off, T.rd16_1 + off);
                    pass_read_tr(aH, TQ, TPL, T.rd16_0 + off, T.rd16_1 + off);
#pragma unroll
                    for (int ta = PROMP_NT - 1; ta >= 0; --ta)
#pragma unroll
                        for (int tb = PROMP_NT - 1 - ta; tb >= 0; --tb) {
                            aw3[c] = mfma16_sw<PROMP_NT>(aR[ta], bD[tb], aw3[c]);
                            aw3[c] = mfma16_sw<PROMP_NT>(aH[ta], bQ[tb], aw3[c]);
                        }
                }
            } else {
                {
                    f32x2 dd, qq;
                    dd[0] = d0;  dd[1] = d1;
                    qq[0] = qm0; qq[1] = qm1;
                    sts2(DB0w, dd);
                    sts2(DB1w, qq);
                }
                wave_fence();
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const float bd = DB0r[t4 * DS], bq = DB1r[t4 * DS];
#pragma unroll
                    for (int c = 0; c < NC2; ++c) aw3[c] = mfma16(TB0r[t4 * TS + 16 * c], bd, aw3[c]);
#pragma unroll
                    for (int c = 0; c < NC2; ++c) aw3[c] = mfma16(TB1r[t4 * TS + 16 * c], bq, aw3[c]);
                }
            }
            CH_TSTAMP(5);
            // ---- dZ2, qZ2 (transposed): ad = W3 dmu^T, aq = W3 qmu^T + (-vW3) dmu^T
            f32x4 dz2[NC2], qz2[NC2];
            {
                f32x2 wb[NC2], vb[NC2];
#pragma unroll
                for (int c = 0; c < NC2; ++c) {
                    wb[c] = lds2(W3b + c * 128);
                    vb[c] = lds2(W3b + VO + c * 128);
                    dz2[c] = zero4();
                    qz2[c] = zero4();
                }
#pragma unroll
                for (int ro = 0; ro < 2; ++ro) {
                    const float dd = ro ? d1 : d0, qq = ro ? qm1 : qm0;
#pragma unroll
                    for (int c = 0; c < NC2; ++c) dz2[c] = mfma16(wb[c][ro], dd, dz2[c]);
#pragma unroll
                    for (int c = 0; c < NC2; ++c) qz2[c] = mfma16(wb[c][ro], qq, qz2[c]);
#pragma unroll
                    for (int c = 0; c < NC2; ++c) qz2[c] = mfma16(vb[c][ro], dd, qz2[c]);
                }
#pragma unroll
                for (int c = 0; c < NC2; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float h = h2[c][r], dd1 = 1.f - h * h, ad = dz2[c][r];
                        qz2[c][r] = qz2[c][r] * dd1 - 2.f * ad * h * rh2[c][r];
                        dz2[c][r] = ad * dd1;
                    }
            }
            CH_TSTAMP(6);
            // ---- out_W2 += R'H1^T dZ2 + H1^T qZ2 ; out_b2 += sum qZ2
            u32x4 dB[NB2][PROMP_NT], qB[NB2][PROMP_NT];     // CACHED: the planes of dZ2 / qZ2 (this product and qZ1 below)
            if (CACHED) {
                // On v_mfma_f32_32x32x16 (K = 16 samples = one tile), float32-equivalent split products.  Two stages: (R'H1, dZ2),
                // (H1, qZ2); the B planes of a stage sit in the full tile TBP, its A planes in TQ.  (Rounds 4-5 walked four stages
                // through a half tile that aliased the float32 transpose tiles; measured equal, the two-stage form is the shorter.)
#pragma unroll
                for (int P = 0; P < NB2; ++P) {
                    pass_split8(dz2[2 * P], dz2[2 * P + 1], dB[P]);
                    pass_split8(qz2[2 * P], qz2[2 * P + 1], qB[P]);
                }
#pragma unroll
                for (int c = 0; c < NC2; ++c) gb2v[c] += qz2[c];
                // (round 6) two stages, each with BOTH 32-unit blocks of its A operand at once: the A operand's planes go through
                // the second full tile (free since the output-kernel gradient read H2 from it) instead of a half tile, block by
                // block -- two LDS round trips per tile instead of four
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    u32x4 aB[NP1][PROMP_NT];
#pragma unroll
                    for (int P = 0; P < NP1; ++P) {
                        if (s == 0) pass_split8(rh1[2 * P], rh1[2 * P + 1], aB[P]);
                        else pass_split8(h1[2 * P], h1[2 * P + 1], aB[P]);
                    }
                    wave_fence();         // the reads of both tiles issued so far (output-kernel gradient / stage 0) precede these writes
#pragma unroll
                    for (int P = 0; P < NB2; ++P) pass_store_planes(TBP, TPL, twr + 256 * P, s == 0 ? dB[P] : qB[P]);
#pragma unroll
                    for (int P = 0; P < NP1; ++P) pass_store_planes(TQ, TPL, twr + 256 * P, aB[P]);
                    wave_fence();
                    u32x4 fa[NB1][PROMP_NT], fb[NB2][PROMP_NT];
#pragma unroll
                    for (int b = 0; b < NB1; ++b) pass_read_tr(fa[b], TQ, TPL, trd0 + 256 * b, trd1 + 256 * b);
#pragma unroll
                    for (int b = 0; b < NB2; ++b) pass_read_tr(fb[b], TBP, TPL, trd0 + 256 * b, trd1 + 256 * b);
#pragma unroll
                    for (int ta = PROMP_NT - 1; ta >= 0; --ta)
#pragma unroll
                        for (int tb = PROMP_NT - 1 - ta; tb >= 0; --tb)
#pragma unroll
                            for (int bi = 0; bi < NB1; ++bi)
#pragma unroll
                                for (int bj = 0; bj < NB2; ++bj) aw2w[bi][bj] = mfma32_sw<PROMP_NT>(fa[bi][ta], fb[bj][tb], aw2w[bi][bj]);
                    sched_fence();
                }
                wave_fence();
            } else {
                wave_fence();
#pragma unroll
                for (int c = 0; c < NC1; ++c) sts4(TB0w + 16 * c, rh1[c]);
#pragma unroll
                for (int c = 0; c < NC2; ++c) sts4(TB1w + 16 * c, dz2[c]);
                wave_fence();
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    float aop[NC1], bop[NC2];
#pragma unroll
                    for (int c = 0; c < NC1; ++c) aop[c] = TB0r[t4 * TS + 16 * c];
#pragma unroll
                    for (int c = 0; c < NC2; ++c) bop[c] = TB1r[t4 * TS + 16 * c];
#pragma unroll
                    for (int i = 0; i < NC1; ++i)
#pragma unroll
                        for (int j = 0; j < NC2; ++j) aw2[i][j] = mfma16(aop[i], bop[j], aw2[i][j]);
                }
                wave_fence();
#pragma unroll
                for (int c = 0; c < NC1; ++c) sts4(TB0w + 16 * c, h1[c]);
#pragma unroll
                for (int c = 0; c < NC2; ++c) sts4(TB1w + 16 * c, qz2[c]);
                wave_fence();
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    float aop[NC1], bop[NC2];
#pragma unroll
                    for (int c = 0; c < NC1; ++c) aop[c] = TB0r[t4 * TS + 16 * c];
#pragma unroll
                    for (int c = 0; c < NC2; ++c) {
                        bop[c] = TB1r[t4 * TS + 16 * c];
                        ob2acc[c] += bop[c];
                    }
#pragma unroll
                    for (int i = 0; i < NC1; ++i)
#pragma unroll
                        for (int j = 0; j < NC2; ++j) aw2[i][j] = mfma16(aop[i], bop[j], aw2[i][j]);
                }
            }
            CH_TSTAMP(7);
            float xN[NOB][4];
            if (!CACHED) chain_load_xN<NOB>(xN, a.obs, base, nrows, O, i16, kk);           // (needed after the next 48 NC1 NC2 MFMAs)
            if (CACHED) {         // the next tile's cache block (the registers' current contents were copied out at the top)
                const int tn = t + NW;
                const long long o = (long long)(tn < tend ? 16 * tn : 16 * seg.tile0) * HCR;
#pragma unroll
                for (int c = 0; c < NC1; ++c) ch1[c] = *(const f32x4*)(hcl + o + 256 * c);
#pragma unroll
                for (int c = 0; c < NC2; ++c) ch2[c] = *(const f32x4*)(hcl + o + 256 * (NC1 + c));
                cmu = *(const f32x2*)(hcm + o);
            }
            // ---- qZ1 (transposed): ad = W2 dZ2^T, aq = W2 qZ2^T + (-vW2) dZ2^T
            f32x4 qz1[NC1];
            {
                f32x4 ad1[NC1];
#pragma unroll
                for (int c = 0; c < NC1; ++c) {
                    ad1[c] = zero4();
                    qz1[c] = zero4();
                }
                if (CACHED) {
                    // On the BF16 pipe (round 3): K = 32 hidden_1 output units per instruction = two 16-unit blocks, a lane's eight
                    // k-slots are its own registers of the two blocks, split three ways; the planes of the second orientation
                    // come from LDS (chain_layout, bwdp).  6 of the 9 term products, smallest first, as in layer 2.
                    chain_gemm2_bf16<NC1, NC2 / 2, false>(qz1, ad1, (const u32x4*)(sm + L.bplanes) + lane,
                                                          (const u32x4*)(sm + L.bplanes + L.bplane_stride) + lane, qB, dB);
                } else {
                // the operands of k-group g + 1 are requested before the products of group g are issued (a fence per group keeps
                // that order): left alone the compiler reads each operand right in front of its product, and the single wave of
                // a SIMD then sits out one LDS latency per product
                float wb[2][NC1], vb[2][NC1];
#pragma unroll
                for (int c1 = 0; c1 < NC1; ++c1) {
                    wb[0][c1] = W2b[c1 * PROMP_CH_BLK];
                    vb[0][c1] = W2b[VO + c1 * PROMP_CH_BLK];
                }
#pragma unroll
                for (int g = 0; g < 4 * NC2; ++g) {
                    const int c2 = g >> 2, r = g & 3, cur = g & 1, nxt = cur ^ 1;
                    sched_fence();
                    if (g + 1 < 4 * NC2) {
                        const int c2n = (g + 1) >> 2, rn = (g + 1) & 3;
#pragma unroll
                        for (int c1 = 0; c1 < NC1; ++c1) {
                            wb[nxt][c1] = W2b[(c2n * NC1 + c1) * PROMP_CH_BLK + 4 * rn];
                            vb[nxt][c1] = W2b[VO + (c2n * NC1 + c1) * PROMP_CH_BLK + 4 * rn];
                        }
                    }
#pragma unroll
                    for (int c1 = 0; c1 < NC1; ++c1) ad1[c1] = mfma16(wb[cur][c1], dz2[c2][r], ad1[c1]);
#pragma unroll
                    for (int c1 = 0; c1 < NC1; ++c1) qz1[c1] = mfma16(wb[cur][c1], qz2[c2][r], qz1[c1]);
#pragma unroll
                    for (int c1 = 0; c1 < NC1; ++c1) qz1[c1] = mfma16(vb[cur][c1], dz2[c2][r], qz1[c1]);
                    if (g + 1 < 4 * NC2) PROMP_SCHED_DSREAD(2 * NC1);      // the requests first, then the products
                    PROMP_SCHED_MFMA(3 * NC1);
                }
                sched_fence();
                }
#pragma unroll
                for (int c = 0; c < NC1; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float h = h1[c][r];
                        const float ad2 = CACHED ? (2.f * cs) * cad1[c][r] : 2.f * ad1[c][r];      // (the cache holds it unscaled)
                        qz1[c][r] = qz1[c][r] * (1.f - h * h) - ad2 * h * rh1[c][r];
                    }
            }
            CH_TSTAMP(8);
            // ---- out_W1 += X^T qZ1 ; out_b1 += sum qZ1
            if (CACHED) {
                // (round 6) on the split pipe, k_pass's hidden_0 kernel gradient: qZ1's planes through the full tile (the hidden_1
                // kernel gradient's reads of it are done), the observations' planes from the tile layer 1 wrote, 32 x 32 x 16 over
                // the 16 samples; the bias gradient from the registers (summed over the sample lanes at the end of the segment)
#pragma unroll
                for (int c = 0; c < NC1; ++c) gb1v[c] += qz1[c];
                u32x4 qB1[NP1][PROMP_NT];
#pragma unroll
                for (int P = 0; P < NP1; ++P) pass_split8(qz1[2 * P], qz1[2 * P + 1], qB1[P]);
                wave_fence();
#pragma unroll
                for (int P = 0; P < NP1; ++P) pass_store_planes(TBP, TPL, twr + 256 * P, qB1[P]);
                wave_fence();
                u32x4 fx[PROMP_NT], fd[NB1][PROMP_NT];
                pass_read_tr(fx, XT, XPL, trd0, trd1);
#pragma unroll
                for (int b = 0; b < NB1; ++b) pass_read_tr(fd[b], TBP, TPL, trd0 + 256 * b, trd1 + 256 * b);
#pragma unroll
                for (int ta = PROMP_NT - 1; ta >= 0; --ta)
#pragma unroll
                    for (int tb = PROMP_NT - 1 - ta; tb >= 0; --tb)
#pragma unroll
                        for (int bj = 0; bj < NB1; ++bj) aw1w[bj] = mfma32_sw<PROMP_NT>(fx[ta], fd[bj][tb], aw1w[bj]);
            } else {
                wave_fence();
#pragma unroll
                for (int c = 0; c < NC1; ++c) sts4(TB0w + 16 * c, qz1[c]);
                wave_fence();
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    float bop[NC1];
#pragma unroll
                    for (int c = 0; c < NC1; ++c) {
                        bop[c] = TB0r[t4 * TS + 16 * c];
                        ob1acc[c] += bop[c];
                    }
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                        for (int c = 0; c < NC1; ++c) aw1[ob][c] = mfma16(xN[ob][t4], bop[c], aw1[ob][c]);
                }
            }
            CH_TSTAMP(9);
        }
        CH_STAMP(2);

        outs0 = row16_sum(outs0);  outs1 = row16_sum(outs1);  outb30 = row16_sum(outb30);
        outb31 = row16_sum(outb31);  klsum = row16_sum(klsum);
#pragma unroll
        for (int j = 0; j < NC1; ++j) ob1acc[j] = fold_groups16(ob1acc[j]);
#pragma unroll
        for (int j = 0; j < NC2; ++j) ob2acc[j] = fold_groups16(ob2acc[j]);
        if (CACHED) {
#pragma unroll
            for (int j = 0; j < NC2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) gb2v[j][r] = row16_sum(gb2v[j][r]);
#pragma unroll
            for (int j = 0; j < NC1; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) gb1v[j][r] = row16_sum(gb1v[j][r]);
        }
        // FP16 split: everything this pass accumulates carries the wave's cs vs; an overflow anywhere in the tile walk ends as an
        // infinity or a NaN in the hidden_0 kernel sums (the last link of every chain): x * 0 is 0 for finite x only.  The waves vote
        // inside chain_reduce_to_partial; a workgroup with an overflow walks the segment again with both scales lowered.
        const float us = PROMP_NT == 2 ? 1.f / qs : 1.f;
        int bad = 0;
        if (PROMP_NT == 2 && attempt + 1 < CHAIN_ATTEMPTS) {
            float chk = 0.f;
            if (CACHED) {
#pragma unroll
                for (int i = 0; i < NB1; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) chk = __builtin_fmaf(aw1w[i][r], 0.f, chk);
            } else {
#pragma unroll
                for (int i = 0; i < NOB; ++i)
#pragma unroll
                    for (int j = 0; j < NC1; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) chk = __builtin_fmaf(aw1[i][j][r], 0.f, chk);
            }
            redo_amax = wave_absmax_f32(amax);
            // (every split of this kernel feeds the hidden_0 kernel sums, except -- cached instance -- the planes of the mean's
            //  cotangents, which feed the output kernel's sums only: their largest values at the scale; amax holds the larger of
            //  |d| and 2^-CHAIN_Q_OVER_D |q|)
            bad = (wave_any(chk != chk) || (CACHED && !(redo_amax * cs * (float)(1 << CHAIN_Q_OVER_D) <= 65504.f))) ? 1 : 0;
        }
        outs0 *= dist[CH_LMASK + q0] * us;
        outs1 *= dist[CH_LMASK + q1] * us;
        float* P = a.partials + (long long)sg * a.partial_stride;
        const bool redo = chain_reduce_to_partial<NC1, NC2, NOB, NW, CACHED>(sm + 4, P, aw2, aw2w, aw1, aw1w, gb1v, aw3, ob1acc, ob2acc, gb2v, outs0, outs1,
                                                                             outb30 * us, outb31 * us, 0.f, klsum, O, A, tid, us, bad, us * w1w);
        if (redo) {               // (the partial row just written is written again)
            if (tid == 0) atomic_add_agent(a.split_events + 1, 1);
            attempt += 1;
            sg -= 1;
            continue;
        }
        attempt = 0;
        CH_STAMP(3);
        if (a.fuse_reduce) chain_task_reduce<NT>(a, (int*)sm, task, NP, tid);
        CH_STAMP(4);
        CH_WGSTAMP(1 + (sg - sg0 < 2 ? sg - sg0 : 1));
    }
}
