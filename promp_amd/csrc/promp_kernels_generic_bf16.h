// promp_kernels_generic_bf16.h -- the GEMMs of the layer-by-layer kernels (promp_kernels_generic.h) on the BF16 matrix pipe.
//
// Same launches, same arguments, same activations / cotangents in global memory (float32), same partial rows as k_gen_linear /
// k_gen_wgrad; what changes is the product: every float32 operand is split into three BF16 terms (x = x0 + x1 + x2 up to
// 2^-24 |x|) and 6 of the 9 cross products are accumulated in float32 by v_mfma_f32_32x32x16_bf16 -- the float32-equivalent
// form of k_pass / k_wb_* (measured error at the level of the exact-FP32 chain).  On this chip the FP32 matrix
// instruction issues at the vector rate and does not overlap vector work, so in k_gen_linear every address
// computation and every LDS store of the staging ADDS to the products' 4096 cycles per 64 x 32 x 256 chunk (measured 7 k cycles
// per chunk at two waves per SIMD, 17 k at one); here the same chunk is 48 matrix instructions per wave (1536 cycles) and the
// vector work runs beside them.
//
//   * Weights: split ONCE per pass by k_gb_planes into a plane copy in global memory (per task when the parameters are), in
//     both orientations and already in the chunk order the GEMMs stage:
//         F  (forward, contraction over the layer's inputs k)    [chunk = k / 32][plane][column n, padded to 64][32 k]
//         B  (backward, contraction over the layer's outputs n)  [chunk = n / 32][plane][column k, padded to 64][32 n]
//     zero-padded, so that a chunk of the weight operand is ONE contiguous block copied 16 bytes per lane into LDS -- no
//     address arithmetic per element, no bounds checks, no split in the GEMM.
//   * Activations / cotangents: float32 rows from global memory, 8 consecutive contraction entries per thread, split in
//     registers on the way into LDS (44 vector instructions per thread and chunk).
//   * LDS tiles: [plane][row or column][32 contraction entries] bf16 with 80-byte rows (64 + 16): the 16-byte slot of row r,
//     part c is 5 r + c -- sixteen consecutive rows hit sixteen different slots, so the b128 fragment reads and the b128
//     stores of the staging are conflict-free (two-way on 3 of 16 lanes for the weight copy).
//   * Software pipeline: the global loads of the next chunk are issued right after the current chunk went into LDS and travel
//     under its matrix instructions (raw values in registers; masks and splits are applied at the store).
//   * The weight gradient contracts over ROWS: its operands are staged transposed ([unit][32 rows], 8 rows per thread through
//     strided loads that are coalesced across the lanes), so that the fragments are the same b128 reads.
#pragma once
#include "promp_kernels_generic.h"
#include <type_traits>

#define GB_R 64                 // rows per round of k_gb_linear / input units per slab of k_gb_wgrad
#define GB_KC 32                // contraction entries per chunk
#define GB_ROWB 80              // bytes per tile row
#define GB_TILE64 (3 * 64 * GB_ROWB)      // bytes of a 64-row operand tile (three planes)
PROMP_HD int gb_up(int x, int m) { return m * ((x + m - 1) / m); }
// elements (16-bit) of one layer's plane blocks
PROMP_HD long long gb_f_elems(int K, int N) { return (long long)gb_up(K, 32) * 3 * gb_up(N, 64); }
PROMP_HD long long gb_b_elems(int K, int N) { return (long long)gb_up(N, 32) * 3 * gb_up(K, 64); }
PROMP_HD int Ly_K_slabs(int K) { return (K + 63) / 64; }
PROMP_HD size_t gb_smem(int nt, int nbw) { return (size_t)nt * (GB_TILE64 + 3 * 64 * nbw * GB_ROWB); }

// developer tooling: cycles per phase of workgroup (0, 0) / thread 0 (-DPROMP_DEV_STAMPS), printed at the end of the launch
#ifdef PROMP_DEV_STAMPS
#define GB_PHASE_DECL unsigned long long gb_ph[6] = {0, 0, 0, 0, 0, 0}, gb_t = promp_clock()
#define GB_PHASE(i) do { const unsigned long long now_ = promp_clock(); gb_ph[i] += now_ - gb_t; gb_t = now_; } while (0)
#else
#define GB_PHASE_DECL do { } while (0)
#define GB_PHASE(i) do { } while (0)
#endif

struct GbPlaneArgs {
    const float* src;               // the parameters [Theta] / [tasks][Theta], or the direction [tasks][Theta]
    long long src_task_stride;
    unsigned short* dst;            // [vectors][plane_stride]
    long long dst_task_stride;
    float sign;                     // -1 for the direction (the R-operator passes run along u = -v)
    int n_lin;
    GenLin lin[GEN_MAX_LIN];
    int pf_off[GEN_MAX_LIN], pb_off[GEN_MAX_LIN];
};

// k_gb_planes: every layer's kernel, split, into the F and B plane blocks.  One thread per (column, 8 contraction entries); the
// lanes run along whichever of the two is contiguous in the kernel's [K][N] storage (F: the column n; B: the contraction index n).
// grid = (x, 2 n_lin, vectors), block = 256.
__global__ void __launch_bounds__(256) k_gb_planes(GbPlaneArgs a) {
    const int li = (int)blockIdx.y >> 1, orient = (int)blockIdx.y & 1;
    const GenLin Ly = a.lin[li];
    const float* W = a.src + (long long)blockIdx.z * a.src_task_stride + Ly.w_off;
    unsigned short* dst = a.dst + (long long)blockIdx.z * a.dst_task_stride + (orient ? a.pb_off[li] : a.pf_off[li]);
    const int K = Ly.K, N = Ly.N;
    const int ncol = orient ? gb_up(K, 64) : gb_up(N, 64);           // the GEMM's output units
    const int nct = orient ? gb_up(N, 32) : gb_up(K, 32);            // contraction length
    const int total = ncol * (nct >> 3);
    for (int q = (int)blockIdx.x * 256 + (int)threadIdx.x; q < total; q += 256 * (int)gridDim.x) {
        const int noc = nct >> 3;
        const int oc = orient ? q % noc : q / ncol, col = orient ? q / noc : q - oc * ncol;
        unsigned w[3][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ct = 8 * oc + 2 * i + h;
                const bool ok = orient ? (ct < N && col < K) : (ct < K && col < N);
                const long long o = orient ? (long long)col * N + ct : (long long)ct * N + col;
                const float wv = W[ok ? o : 0];          // (unconditional: a load under `ok ? :` is a branch and a wait of its own -- eight serial round trips per thread)
                x[h] = ok ? a.sign * wv : 0.f;
            }
            unsigned t[3];
            bf16_split3_pair(x[0], x[1], t);
            w[0][i] = t[0];
            w[1][i] = t[1];
            w[2][i] = t[2];
        }
        const int chunk = oc >> 2, part = oc & 3;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            u32x4 v;
            v[0] = w[p][0]; v[1] = w[p][1]; v[2] = w[p][2]; v[3] = w[p][3];
            *(u32x4*)(dst + ((long long)(chunk * 3 + p) * ncol + col) * 32 + 8 * part) = v;
        }
    }
}

// the six products of a float32-equivalent 32 x 32 x 16 block, small terms first
PROMP_DEV void gb_mma6(f32x16& c, const u32x4 (&x)[3], const u32x4 (&y)[3]) {
    c = mfma32_bf16w(x[2], y[0], c);
    c = mfma32_bf16w(x[1], y[1], c);
    c = mfma32_bf16w(x[0], y[2], c);
    c = mfma32_bf16w(x[1], y[0], c);
    c = mfma32_bf16w(x[0], y[1], c);
    c = mfma32_bf16w(x[0], y[0], c);
}
// eight float32 values -> three planes of eight bf16, 16 bytes each, at tile + 3 planes of `rows` rows
PROMP_DEV void gb_split_store(const float (&x)[8], unsigned char* tile, int rows, int row, int part) {
    u32x4 v[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned t[3];
        bf16_split3_pair(x[2 * i], x[2 * i + 1], t);
        v[0][i] = t[0];
        v[1][i] = t[1];
        v[2][i] = t[2];
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) *(u32x4*)(tile + ((size_t)p * rows + row) * GB_ROWB + 16 * part) = v[p];
}
PROMP_DEV void gb_frag(u32x4 (&f)[3], const unsigned char* tile, int rows, int row, int ks, int hi) {
#pragma unroll
    for (int p = 0; p < 3; ++p) f[p] = *(const u32x4*)(tile + ((size_t)p * rows + row) * GB_ROWB + 32 * ks + 16 * hi);
}

// k_gb_linear: k_gen_linear's four modes (see there for the mathematics) with the products on the BF16 pipe.
// 64 rows x 64 NBW columns per round; 32-column blocks: wave w owns blocks w, w + 4 (both 32-row blocks), at NBW = 1 one block of one
// row block.  grid = (work items, GEN_SPLIT), block = 256, smem = gb_smem(1 or 2, NBW).
template <int MODE, int NBW>
__global__ void __launch_bounds__(256) k_gb_linear(GenArgs a, int li, int pp) {
    PROMP_SMEM_DECL;
    constexpr bool TAN = MODE == GEN_FWD_T || MODE == GEN_BWD_T;
    constexpr bool FWD = MODE == GEN_FWD || MODE == GEN_FWD_T;
    constexpr int NC = 64 * NBW, NCB = 2 * NBW;
    constexpr int CPW = NBW == 1 ? 1 : (NCB + 3) / 4, RPW = NBW == 1 ? 1 : 2;
    constexpr int NPB = 12 * NC / 256;                       // 16-byte pieces of a weight chunk per thread
    unsigned char* As = PROMP_SMEM_PTR;                       // [3][64][80 B]
    unsigned char* RAs = As + GB_TILE64;                      // (TAN)
    unsigned char* Bs = As + (TAN ? 2 : 1) * GB_TILE64;       // [3][NC][80 B]
    unsigned char* Us = Bs + 3 * NC * GB_ROWB;                // (TAN) minus the direction's kernel
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    // Consecutive workgroups go to consecutive XCDs (8, each with its own 4 MB L2), and the work items are ordered by task: with
    // item = blockIdx.x every XCD streams every task's planes (40 x 393 KB for a 256 x 256 layer).  Every XCD gets a contiguous
    // eighth of the items instead -- five tasks' planes (-3 % per pass at 256 x 256.  Going further -- a 1-D grid in which an XCD
    // walks its items one or two tasks at a time -- measured slower, 15.9 against 13.2 ms per step, with unchanged phase stamps:
    // the 24 B / clock / CU at which a chunk's 28 loads issue is not an L2 miss rate).
    const int per = (int)gridDim.x >> 3, bx = (int)blockIdx.x;
    const int split = (int)blockIdx.y;
    const WorkItem wk = a.work[bx < 8 * per ? (bx & 7) * per + (bx >> 3) : bx];
    const GenLin Ly = a.lin[li];
    const float* th = a.theta + (long long)wk.task * a.theta_task_stride;
    const bool last = li == a.n_lin - 1;
    const int Kc = FWD ? Ly.K : Ly.N;            // contraction length
    const int Nc = FWD ? Ly.N : Ly.K;            // output width
    const float* __restrict__ Ain = FWD ? a.act[li] : a.dz[pp];                       // rows of width Kc
    const float* __restrict__ RAin = !TAN ? nullptr : FWD ? (li > 0 ? a.ract[li] : nullptr) : a.qz[pp];
    const int poff = FWD ? a.pf_off[li] : a.pb_off[li];
    const u32x4* __restrict__ Wp = (const u32x4*)(a.wplanes + (long long)wk.task * a.wplane_stride + poff);
    const u32x4* __restrict__ Up = TAN ? (const u32x4*)(a.vplanes + (long long)wk.task * a.vplane_stride + poff) : nullptr;
    // 16-byte loads of the rows where the width and the base allow them (wave-uniform)
    const bool vec = (Kc & 3) == 0 && (((size_t)Ain) & 15) == 0 && (RAin == nullptr || (((size_t)RAin) & 15) == 0);
    const int rb0 = NBW == 1 ? (w >> 1) : 0;
    const int sr = tid >> 2, so = tid & 3;       // staging: row, 8-entry part
    // The requested chunk of activations (and of their tangents) stays in four VECTOR values across the products, not in float
    // arrays: with arrays the scalar-load path below kept a runtime-indexed store, the arrays then lived partly in scratch memory, and
    // in the tangent modes every 16-byte request was waited for (s_waitcnt vmcnt(0)) and copied to scratch on the spot -- four
    // serial round trips to memory per request instead of a prefetch under the products.
    f32x4 xa0 = {0.f, 0.f, 0.f, 0.f}, xa1 = xa0, xra0 = xa0, xra1 = xa0;
    u32x4 pb[NPB], pu[NPB];
    const int rstep = GB_R * (int)gridDim.y;
    auto issue = [&](int row0, int k0) {
        const int nrows = wk.row_end - row0 < GB_R ? wk.row_end - row0 : GB_R;
        const long long ro = (long long)(row0 + (sr < nrows ? sr : nrows - 1)) * Kc;
        const int k = k0 + 8 * so;
        if (vec) {
            const int kq0 = k < Kc ? k : 0, kq1 = k + 4 < Kc ? k + 4 : 0;
            xa0 = *(const f32x4*)(Ain + ro + kq0);
            xa1 = *(const f32x4*)(Ain + ro + kq1);
            if (TAN && RAin != nullptr) {
                xra0 = *(const f32x4*)(RAin + ro + kq0);
                xra1 = *(const f32x4*)(RAin + ro + kq1);
            }
        } else {
            const float* pa_ = Ain + ro;
            const int km = Kc - 1;
#define GB_KQ(i) (k + (i) < Kc ? k + (i) : km)
            xa0 = f32x4{pa_[GB_KQ(0)], pa_[GB_KQ(1)], pa_[GB_KQ(2)], pa_[GB_KQ(3)]};
            xa1 = f32x4{pa_[GB_KQ(4)], pa_[GB_KQ(5)], pa_[GB_KQ(6)], pa_[GB_KQ(7)]};
            if (TAN && RAin != nullptr) {
                const float* pr_ = RAin + ro;
                xra0 = f32x4{pr_[GB_KQ(0)], pr_[GB_KQ(1)], pr_[GB_KQ(2)], pr_[GB_KQ(3)]};
                xra1 = f32x4{pr_[GB_KQ(4)], pr_[GB_KQ(5)], pr_[GB_KQ(6)], pr_[GB_KQ(7)]};
            }
#undef GB_KQ
        }
        const long long cb = (long long)(k0 >> 5) * (12 * NC);
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            pb[i] = Wp[cb + tid + 256 * i];
            if (TAN) pu[i] = Up[cb + tid + 256 * i];
        }
    };
    // (the first chunk of the workgroup's NEXT round is requested under the last products of this one)
    GB_PHASE_DECL;
    const int rfirst = wk.row_begin + GB_R * split;
    if (rfirst < wk.row_end) issue(rfirst, 0);
    GB_PHASE(0);
    for (int row0 = rfirst; row0 < wk.row_end; row0 += rstep) {
        const int nrows = wk.row_end - row0 < GB_R ? wk.row_end - row0 : GB_R;
        f32x16 acc[RPW][CPW], racc[RPW][CPW];
#pragma unroll
        for (int rb = 0; rb < RPW; ++rb)
#pragma unroll
            for (int c = 0; c < CPW; ++c)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[rb][c][j] = racc[rb][c][j] = 0.f;
        for (int k0 = 0; k0 < Kc; k0 += GB_KC) {
            __syncthreads();                 // the previous chunk's products are done with the tiles
            GB_PHASE(1);
            {
                float x[8], rx[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool ok = sr < nrows && k0 + 8 * so + i < Kc;
                    x[i] = ok ? (i < 4 ? xa0[i & 3] : xa1[i & 3]) : 0.f;
                    rx[i] = (TAN && ok && RAin != nullptr) ? (i < 4 ? xra0[i & 3] : xra1[i & 3]) : 0.f;
                }
                gb_split_store(x, As, 64, sr, so);
                if (TAN) gb_split_store(rx, RAs, 64, sr, so);
#pragma unroll
                for (int i = 0; i < NPB; ++i) {
                    const int q = tid + 256 * i, p = q / (4 * NC), rem = q - p * (4 * NC);
                    const size_t o = ((size_t)p * NC + (rem >> 2)) * GB_ROWB + 16 * (rem & 3);
                    *(u32x4*)(Bs + o) = pb[i];
                    if (TAN) *(u32x4*)(Us + o) = pu[i];
                }
            }
            __syncthreads();
            GB_PHASE(2);
            {
                const bool more = k0 + GB_KC < Kc;
                const int nrow0 = more ? row0 : row0 + rstep;
                if (nrow0 < wk.row_end) issue(nrow0, more ? k0 + GB_KC : 0);
            }
            GB_PHASE(3);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (k0 + 16 * ks < Kc) {       // (uniform)
                    u32x4 fa[RPW][3], fra[RPW][3];
#pragma unroll
                    for (int rb = 0; rb < RPW; ++rb) {
                        gb_frag(fa[rb], As, 64, 32 * (rb0 + rb) + l31, ks, hi);
                        if (TAN) gb_frag(fra[rb], RAs, 64, 32 * (rb0 + rb) + l31, ks, hi);
                    }
#pragma unroll
                    for (int c = 0; c < CPW; ++c) {
                        const int cbi = NBW == 1 ? (w & 1) : w + 4 * c;
                        if (cbi < NCB && 32 * cbi < Nc) {      // (wave-uniform)
                            u32x4 fb[3], fu[3];
                            gb_frag(fb, Bs, NC, 32 * cbi + l31, ks, hi);
                            if (TAN) gb_frag(fu, Us, NC, 32 * cbi + l31, ks, hi);
#pragma unroll
                            for (int rb = 0; rb < RPW; ++rb) {
                                gb_mma6(acc[rb][c], fa[rb], fb);
                                if (TAN) {
                                    gb_mma6(racc[rb][c], fa[rb], fu);
                                    gb_mma6(racc[rb][c], fra[rb], fb);
                                }
                            }
                        }
                    }
                }
            }
        }
        // D: col = l31, row = (j & 3) + 8 (j >> 2) + 4 hi.  One 64-bit address per output array and round; the 16 rows of a block are
        // 32-bit offsets from it.  BWD: the block's 16 activations (and tangents) are requested together BEFORE the first store
        // (the output arrays may alias them as far as the compiler knows -- which also keeps it from hoisting the NEXT block's
        // loads over this block's stores, 128 more live registers --: load / store pairs in one loop ran one memory round trip after
        // the other, 90 us per launch at 256 x 256).  The nonlinearity is a compile-time constant inside the loops.
        GB_PHASE(4);
        const int kind = (FWD && last) ? gen_out(a.act_kind) : gen_hidden(a.act_kind);
        auto epilogue = [&](auto kind_c) {
            constexpr int KIND = decltype(kind_c)::value;
            const int hz = hi + opaque_zero();      // (per round: keeps the block offsets below from being hoisted out of the row loop and spilled)
            const long long rbase = (long long)(row0 + 4 * hz) * Nc;
            const float* Hp = FWD ? nullptr : a.act[li] + rbase;     // BWD: this layer's input = the previous hidden layer's output
            const float* RHp = (FWD || !TAN) ? nullptr : a.ract[li] + rbase;
            float* O1 = (FWD ? (last ? a.mu : a.out_act[li + 1]) : a.dz[pp ^ 1]) + rbase;
            float* O2 = !TAN ? nullptr : (FWD ? (last ? a.rmu : a.ract[li + 1]) : a.qz[pp ^ 1]) + rbase;
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const int cbi = NBW == 1 ? (w & 1) : w + 4 * c;
                const int col = 32 * cbi + l31 + (hz - hi);
                if (cbi < NCB && col < Nc) {
                    const float b = FWD ? th[Ly.b_off + col] : 0.f;
                    const float ub = (FWD && TAN) ? -a.vdir[(long long)wk.task * a.NP + Ly.b_off + col] : 0.f;
#pragma unroll
                    for (int rb = 0; rb < RPW; ++rb) {
                        const int r0 = 32 * (rb0 + rb) + 4 * hz;           // (this lane's first row of the block; rbase already has 4 hi)
                        const int nleft = nrows - r0;                      // rows (j & 3) + 8 (j >> 2) < nleft are inside
                        const int o0 = 32 * (rb0 + rb) * Nc + col;
                        float hv[16], rhv[16];
                        if (!FWD) {
#pragma unroll
                            for (int j = 0; j < 16; ++j) {
                                const int rj = (j & 3) + 8 * (j >> 2);
                                const int o = rj < nleft ? o0 + rj * Nc : col - 4 * hz * Nc;     // (outside: row0's entry)
                                hv[j] = Hp[o];
                                if (TAN) rhv[j] = RHp[o];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const int rj = (j & 3) + 8 * (j >> 2);
                            if (rj < nleft) {
                                const int o = o0 + rj * Nc;
                                if (FWD) {
                                    const float h = gen_act(KIND, acc[rb][c][j] + b);
                                    O1[o] = h;
                                    if (TAN) O2[o] = gen_act_d(KIND, h) * (racc[rb][c][j] + ub);
                                } else {
                                    const float h = hv[j], d1 = gen_act_d(KIND, h), dx = acc[rb][c][j];
                                    O1[o] = dx * d1;
                                    if (TAN) O2[o] = racc[rb][c][j] * d1 - (KIND == GEN_ACT_TANH ? 2.f * dx * h * rhv[j] : 0.f);
                                }
                            }
                        }
                    }
                }
            }
        };
        if (kind == GEN_ACT_TANH) epilogue(std::integral_constant<int, GEN_ACT_TANH>{});
        else if (kind == GEN_ACT_RELU) epilogue(std::integral_constant<int, GEN_ACT_RELU>{});
        else epilogue(std::integral_constant<int, GEN_ACT_IDENTITY>{});
        GB_PHASE(5);
    }
#ifdef PROMP_DEV_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
        printf("k_gb_linear<%d,%d> li %d Kc %d Nc %d cycles: first issue %llu | wait for products %llu | split + store %llu | issue next %llu | products %llu | epilogue %llu\n",
               MODE, NBW, li, Kc, Nc, gb_ph[0], gb_ph[1], gb_ph[2], gb_ph[3], gb_ph[4], gb_ph[5]);
#endif
}

// k_gb_wgrad: k_gen_wgrad (this work item's share of a layer's kernel / bias gradient into its partial row) on the BF16 pipe.
// Output slab: 64 input units (blockIdx.y) x 64 NBW output units, summed over the work item's rows 32 at a time; both operands are
// staged transposed ([unit][32 rows]).  Sums run in row order inside a workgroup: bitwise reproducible.
// grid = work items x ceil(K / 64), block = 256, smem = gb_smem(NT, NBW).
template <int NT, int NBW>
__global__ void __launch_bounds__(256) k_gb_wgrad(GenArgs a, int li, int pp) {
    PROMP_SMEM_DECL;
    constexpr int NC = 64 * NBW, NCB = 2 * NBW;
    constexpr int CPW = NBW == 1 ? 1 : (NCB + 3) / 4, RPW = NBW == 1 ? 1 : 2;
    unsigned char* Xs = PROMP_SMEM_PTR;                       // [3][64 input units][80 B]
    unsigned char* RXs = Xs + GB_TILE64;                      // (NT == 2)
    unsigned char* Ds = Xs + NT * GB_TILE64;                  // [3][NC output units][80 B]
    unsigned char* Qs = Ds + 3 * NC * GB_ROWB;                // (NT == 2)
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    // 1-D grid of work items x slabs.  Consecutive workgroups go to consecutive XCDs (8, each with its own L2): workgroup id = 8 t + x
    // is slab t % nslab of work item 8 (t / nslab) + x, so that the slabs of an item -- which all read the item's cotangent rows, 4 x
    // 82 MB per launch at 256 x 256 -- are dispatched back to back onto ONE XCD and share the rows through its L2.
    const int nslab = (Ly_K_slabs(a.lin[li].K)), nitem = (int)gridDim.x / nslab, n8 = (nitem >> 3) * 8;
    int item, slab;
    {
        const int id = (int)blockIdx.x;
        if (id < n8 * nslab) { const int t = id >> 3; slab = t % nslab; item = 8 * (t / nslab) + (id & 7); }
        else { const int r = id - n8 * nslab; item = n8 + r / nslab; slab = r - (r / nslab) * nslab; }
    }
    const WorkItem wk = a.work[item];
    const GenLin Ly = a.lin[li];
    const int K = Ly.K, N = Ly.N;
    float* P = a.partials + (long long)item * a.partial_stride;
    const float* __restrict__ X = a.act[li];
    const float* __restrict__ RX = (NT == 2 && li > 0) ? a.ract[li] : nullptr;
    const float* __restrict__ DZ = a.dz[pp];
    const float* __restrict__ QZ = NT == 2 ? a.qz[pp] : nullptr;
    const int rb0 = NBW == 1 ? (w >> 1) : 0;
    const int su = lane, so = w;                 // staging: unit (of a block of 64), 8-row part
    {
        const int kb0 = 64 * slab;
        f32x16 acc[RPW][CPW];
#pragma unroll
        for (int rb = 0; rb < RPW; ++rb)
#pragma unroll
            for (int c = 0; c < CPW; ++c)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[rb][c][j] = 0.f;
        float bs[NBW];                                     // bias gradient: this thread's rows of column 64 c + su
#pragma unroll
        for (int c = 0; c < NBW; ++c) bs[c] = 0.f;
        float xx[8], xr[8], xd[NBW][8], xq[NBW][8];
        auto issue = [&](int row0) {
            // (one 64-bit base per array and chunk; the rows are 32-bit offsets from it)
            const int nrows = wk.row_end - row0 < GB_KC ? wk.row_end - row0 : GB_KC;
            const int kq = kb0 + su < K ? kb0 + su : K - 1;
            const float* xb = X + (long long)row0 * K + kq;
            const float* rxb = (NT == 2 && RX != nullptr) ? RX + (long long)row0 * K + kq : nullptr;
            const float* db = DZ + (long long)row0 * N;
            const float* qb = NT == 2 ? QZ + (long long)row0 * N : nullptr;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = 8 * so + i, rr = r < nrows ? r : nrows - 1;
                xx[i] = xb[rr * K];
                if (NT == 2 && RX != nullptr) xr[i] = rxb[rr * K];
#pragma unroll
                for (int c = 0; c < NBW; ++c) {
                    const int n = 64 * c + su < N ? 64 * c + su : N - 1;
                    xd[c][i] = db[rr * N + n];
                    if (NT == 2) xq[c][i] = qb[rr * N + n];
                }
            }
        };
        issue(wk.row_begin);
        for (int row0 = wk.row_begin; row0 < wk.row_end; row0 += GB_KC) {
            const int nrows = wk.row_end - row0 < GB_KC ? wk.row_end - row0 : GB_KC;
            __syncthreads();
            {
                float x[8], y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool ok = 8 * so + i < nrows && kb0 + su < K;
                    x[i] = ok ? xx[i] : 0.f;
                    y[i] = (NT == 2 && ok && RX != nullptr) ? xr[i] : 0.f;
                }
                gb_split_store(x, Xs, 64, su, so);
                if (NT == 2) gb_split_store(y, RXs, 64, su, so);
#pragma unroll
                for (int c = 0; c < NBW; ++c) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const bool ok = 8 * so + i < nrows && 64 * c + su < N;
                        x[i] = ok ? xd[c][i] : 0.f;
                        y[i] = (NT == 2 && ok) ? xq[c][i] : 0.f;
                        bs[c] += NT == 2 ? y[i] : x[i];
                    }
                    gb_split_store(x, Ds, NC, 64 * c + su, so);
                    if (NT == 2) gb_split_store(y, Qs, NC, 64 * c + su, so);
                }
            }
            __syncthreads();
            if (row0 + GB_KC < wk.row_end) issue(row0 + GB_KC);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (16 * ks < nrows) {
                    u32x4 fx[RPW][3], frx[RPW][3];
#pragma unroll
                    for (int rb = 0; rb < RPW; ++rb) {
                        gb_frag(fx[rb], Xs, 64, 32 * (rb0 + rb) + l31, ks, hi);
                        if (NT == 2) gb_frag(frx[rb], RXs, 64, 32 * (rb0 + rb) + l31, ks, hi);
                    }
#pragma unroll
                    for (int c = 0; c < CPW; ++c) {
                        const int cbi = NBW == 1 ? (w & 1) : w + 4 * c;
                        if (cbi < NCB && 32 * cbi < N) {      // (wave-uniform)
                            u32x4 fd[3], fq[3];
                            gb_frag(fd, Ds, NC, 32 * cbi + l31, ks, hi);
                            if (NT == 2) gb_frag(fq, Qs, NC, 32 * cbi + l31, ks, hi);
#pragma unroll
                            for (int rb = 0; rb < RPW; ++rb) {
                                if (NT == 1) gb_mma6(acc[rb][c], fx[rb], fd);
                                else {
                                    gb_mma6(acc[rb][c], frx[rb], fd);
                                    gb_mma6(acc[rb][c], fx[rb], fq);
                                }
                            }
                        }
                    }
                }
            }
        }
        // D: col = l31 (output unit), row = (j & 3) + 8 (j >> 2) + 4 hi (input unit of the 32-block)
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int cbi = NBW == 1 ? (w & 1) : w + 4 * c;
            const int col = 32 * cbi + l31;
            if (cbi < NCB && col < N) {
#pragma unroll
                for (int rb = 0; rb < RPW; ++rb)
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int k = kb0 + 32 * (rb0 + rb) + (j & 3) + 8 * (j >> 2) + 4 * hi;
                        if (k < K) P[Ly.w_off + (long long)k * N + col] = acc[rb][c][j];
                    }
            }
        }
        if (kb0 == 0) {
            // the bias gradient: the four 8-row parts of every column in fixed order (through the tile memory)
            float* scr = (float*)PROMP_SMEM_PTR;
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NBW; ++c) scr[so * NC + 64 * c + su] = bs[c];
            __syncthreads();
            for (int n = tid; n < N; n += 256) P[Ly.b_off + n] = (scr[n] + scr[NC + n]) + (scr[2 * NC + n] + scr[3 * NC + n]);
        }
    }
}
