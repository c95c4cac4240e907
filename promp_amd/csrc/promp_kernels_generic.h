// promp_kernels_generic.h -- the policy passes for ANY tanh-MLP shape the reference's create_mlp can build
// (policies/networks/mlp.py:5-62): 1 .. 4 hidden layers, widths up to 256, obs_dim up to 1024, act_dim up to 64 -- e.g. three
// hidden layers, or the reference's Humanoid environments (376 observations, 17 actions:
// envs/mujoco_envs/humanoid_rand_direc.py:34-41), which the fused kernels (k_pass / k_chain_hvp: two layers of 32 / 64 units,
// obs_dim <= 32; k_wide_*: (64,64) / (128,128), obs_dim <= 128, act_dim <= 8) do not cover.
//
// Same mathematics (oracle/promp.py: loss_and_grad, hvp), same PassArgs semantics, same partial rows (one per work item, summed
// by k_reduce_task) as those kernels -- but layer by layer, one launch per layer and direction, with every layer's activations
// and cotangents in global memory: a pass is a chain of
//     k_gen_linear<FWD | FWD_T>  (per layer)   H_l = tanh(H_{l-1} W_l + b_l)       [+ tangent R'H_l along -v]
//     k_gen_loss / k_gen_loss_hvp               objective, KL, cotangents of the mean [+ their tangents], log_std terms
//     k_gen_wgrad<1 | 2>         (per layer)   dW_l = H_{l-1}^T dZ_l, db_l           [R-operator: R'H^T dZ + H^T qZ]
//     k_gen_linear<BWD | BWD_T>  (per layer)   dZ_{l-1} = (dZ_l W_l^T) (1 - H^2)     [+ qZ_{l-1}]
// It is the general fallback, not the fast path: activations travel through HBM / the Infinity Cache between the launches
// (the fused kernels keep them in registers), the GEMMs are exact-FP32 MFMA (v_mfma_f32_16x16x4_f32) on LDS-staged row tiles.
//
// R-operator convention (as k_chain_hvp): the direction is u = -v, so that the pass yields -H v directly; the KL term of the
// inner objective (kl_weight x grad KL) joins at the mean level.
#pragma once
#include "promp_kernels_policy.h"
#include "promp_kernels_rollout.h"

#define GEN_MAX_LIN 5          // linear layers: up to 4 hidden + the output layer
#define GEN_R 64               // rows per chunk (4 row blocks of 16)
#define GEN_KC 64              // contraction entries staged per chunk
#define GEN_LD (GEN_KC + 16)   // LDS row stride of a staged chunk of k_gen_wgrad (4 rows x 16 columns of an operand read: 64 banks)
#define GEN_RW 32              // rows per chunk of the weight-gradient kernel (its LDS also holds the cotangent rows)
#define GEN_MAX_N 256          // widest layer output (4 waves x 4 column blocks of 16)
#define GEN_MAX_A 64
#define GEN_SPLIT 4            // workgroups per work item of k_gen_linear

struct GenLin {
    int K, N;                  // in / out width
    int w_off, b_off;          // offsets of the kernel [K][N] and the bias [N] in the parameter vector (reference order)
};

struct GenArgs {
    const WorkItem* work;              // one row range of one task per workgroup (table 0)
    const int* task_row_offsets;
    int n_lin;                         // hidden layers + 1
    GenLin lin[GEN_MAX_LIN];
    int O, A, NP;
    const float* theta;                // [Theta] or [tasks][Theta]
    long long theta_task_stride;
    const float* vdir;                 // R-operator passes: [tasks][Theta]
    // activations, row-major: act[0] = observations [rows][O], act[l] = output of hidden layer l [rows][N_l]; ract = tangents
    const float* act[GEN_MAX_LIN];
    float* ract[GEN_MAX_LIN];
    float* out_act[GEN_MAX_LIN];       // (writable aliases of act[1..])
    float *mu, *rmu;                   // [rows][A]
    float *dz[2], *qz[2];              // cotangent ping-pong buffers [rows][<= GEN_MAX_N]
    // row data
    const float *actions, *adv, *old_mean, *old_log_std;
    int ls_per_row;
    float* partials;
    int partial_stride;
    int loss_kind;
    float clip_eps;
    int clip_log_std;
    float min_log_std;
    float kl_weight;
    float* row_tan;
    int act_kind;                      // low byte: hidden nonlinearity, GEN_ACT_TANH / _RELU / _IDENTITY (policies/networks/mlp.py:47 takes any);
                                       // bits 8..: the same code for output_nonlinearity (mlp.py:53-60, 114-117; none = GEN_ACT_IDENTITY): gen_hidden / gen_out
    // the BF16 plane copies of the parameters / of minus the direction (promp_kernels_generic_bf16.h: k_gb_planes), 16-bit elements
    const unsigned short *wplanes, *vplanes;
    long long wplane_stride, vplane_stride;        // per task (0: one copy for all tasks)
    int pf_off[GEN_MAX_LIN], pb_off[GEN_MAX_LIN];  // a layer's forward / backward plane block
};

enum { GEN_FWD = 0, GEN_FWD_T = 1, GEN_BWD = 2, GEN_BWD_T = 3 };
// Hidden nonlinearities.  The derivative is a function of the OUTPUT for all three (tanh: 1 - h^2; relu: h > 0, TF's relu'(0) = 0;
// identity: 1), so the backward kernels need no pre-activations; only tanh has a second derivative (-2 h (1 - h^2)).
enum { GEN_ACT_TANH = 0, GEN_ACT_RELU = 1, GEN_ACT_IDENTITY = 2 };
PROMP_DEV int gen_hidden(int packed) { return packed & 0xff; }
PROMP_DEV int gen_out(int packed) { return packed >> 8; }
PROMP_DEV float gen_act(int kind, float z) { return kind == GEN_ACT_TANH ? fast_tanh(z) : kind == GEN_ACT_RELU ? fmaxf(z, 0.f) : z; }
PROMP_DEV float gen_act_d(int kind, float h) { return kind == GEN_ACT_TANH ? 1.f - h * h : kind == GEN_ACT_RELU ? (h > 0.f ? 1.f : 0.f) : 1.f; }
PROMP_HD int gen_wgrad_ld(int N) { return 64 * ((N + 63) / 64) + 16; }
PROMP_HD size_t gen_wgrad_smem(int nt, int N) { return sizeof(float) * (size_t)nt * GEN_RW * (GEN_LD + gen_wgrad_ld(N)); }

// LDS of k_gen_linear: NA row tiles [GEN_R][GEN_LA] (the layer's input rows / cotangent rows, and their tangents) and NB weight
// tiles [GEN_KL][ncols + 16] (the layer's kernel, and the direction's), one chunk of GEN_KL contraction entries at a time.
// Row strides: GEN_LA = 36 and (64 NBW + 16) put the 64 lanes of an MFMA operand read on 64 different banks.
#define GEN_KL 32
#define GEN_LA (GEN_KL + 4)
PROMP_HD size_t gen_linear_smem(int mode, int nbw) {
    const int nt = (mode == 1 || mode == 3) ? 2 : 1;
    return sizeof(float) * (size_t)nt * (GEN_R * GEN_LA + GEN_KL * (64 * nbw + 16));
}

// k_gen_linear: one linear layer over the work item's rows, forward or backward, primal or primal + tangent.
//   FWD   : H = f(X W + b)                                        f = tanh (hidden) or identity (li == n_lin - 1: the means)
//   FWD_T : ... and R'H = f'(.) (R'X W + X U + ub),  U = -v's slice (R'X = 0 for the observations)
//   BWD   : dZ_prev = (dZ W^T) (1 - H_prev^2)                     (li >= 1: the gradient stops at the observations)
//   BWD_T : ... and qZ_prev = (qZ W^T + dZ U^T) (1 - H_prev^2) - 2 (dZ W^T) H_prev R'H_prev
// Exact-FP32 MFMA (v_mfma_f32_16x16x4_f32) on LDS-staged tiles: 64 rows x the whole output width per round, wave w owns the
// column blocks w, w + 4, ...; the tangent products share the staged tiles with the primal one.
// grid = work items, block = 256, smem = gen_linear_smem(MODE, NBW).  NBW = column blocks per wave (output width <= 64 NBW).
template <int MODE, int NBW>
__global__ void __launch_bounds__(256) k_gen_linear(GenArgs a, int li, int pp) {
    PROMP_SMEM_DECL;
    constexpr bool TAN = MODE == GEN_FWD_T || MODE == GEN_BWD_T;
    constexpr bool FWD = MODE == GEN_FWD || MODE == GEN_FWD_T;
    constexpr int NCS = 64 * NBW + 16;
    float* As = (float*)PROMP_SMEM_PTR;                       // [GEN_R][GEN_LA]
    float* RAs = As + GEN_R * GEN_LA;                         // (TAN)
    float* Bs = As + (TAN ? 2 : 1) * GEN_R * GEN_LA;          // [GEN_KL][NCS]
    float* Us = Bs + GEN_KL * NCS;                            // (TAN) minus the direction's kernel
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6), i16 = lane & 15, kk = lane >> 4;
    const WorkItem wk = a.work[blockIdx.x];
    const GenLin Ly = a.lin[li];
    const float* th = a.theta + (long long)wk.task * a.theta_task_stride;
    const float* W = th + Ly.w_off;
    const float* U = TAN ? a.vdir + (long long)wk.task * a.NP + Ly.w_off : nullptr;
    const bool last = li == a.n_lin - 1;
    const int Kc = FWD ? Ly.K : Ly.N;            // contraction length
    const int Nc = FWD ? Ly.N : Ly.K;            // output width
    const float* Ain = FWD ? a.act[li] : a.dz[pp];                       // rows of width Kc
    const float* RAin = !TAN ? nullptr : FWD ? (li > 0 ? a.ract[li] : nullptr) : a.qz[pp];
    // (the rounds of a work item are dealt to gridDim.y workgroups: more waves per CU to hide the staging latency)
    for (int row0 = wk.row_begin + GEN_R * (int)blockIdx.y; row0 < wk.row_end; row0 += GEN_R * (int)gridDim.y) {
        const int nrows = wk.row_end - row0 < GEN_R ? wk.row_end - row0 : GEN_R;
        f32x4 acc[4][NBW], racc[4][NBW];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int c = 0; c < NBW; ++c) acc[rb][c] = racc[rb][c] = zero4();
        for (int k0 = 0; k0 < Kc; k0 += GEN_KL) {
            // every load of the round is issued before the first wait: clamped (always valid) addresses, zeros selected afterwards
            constexpr int NLA = GEN_R * GEN_KL / 256, NLB = GEN_KL * 64 * NBW / 256;
            float va[NLA], vra[NLA], vb[NLB], vu[NLB];
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                const int e = tid + 256 * i, r = e >> 5, k = e & 31;
                const int rr = r < nrows ? r : nrows - 1, kc = k0 + k < Kc ? k0 + k : Kc - 1;
                const long long o = (long long)(row0 + rr) * Kc + kc;
                va[i] = Ain[o];
                if (TAN) vra[i] = RAin != nullptr ? RAin[o] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < NLB; ++i) {
                const int e = tid + 256 * i;
                long long o;
                if (FWD) {
                    const int k = e / (64 * NBW), n = e - k * (64 * NBW);
                    o = (long long)(k0 + k < Kc ? k0 + k : Kc - 1) * Ly.N + (n < Nc ? n : Nc - 1);
                } else {
                    // contraction over the layer's output units n, output columns = its input units k:  B[n][k] = W[k][n]
                    const int n = e & 31, k = e >> 5;
                    o = (long long)(k < Nc ? k : Nc - 1) * Ly.N + (k0 + n < Kc ? k0 + n : Kc - 1);
                }
                vb[i] = W[o];
                if (TAN) vu[i] = U[o];
            }
            __syncthreads();                 // the previous chunk's products are done with the tiles
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                const int e = tid + 256 * i, r = e >> 5, k = e & 31;
                const bool ok = r < nrows && k0 + k < Kc;
                As[r * GEN_LA + k] = ok ? va[i] : 0.f;
                if (TAN) RAs[r * GEN_LA + k] = ok ? vra[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < NLB; ++i) {
                const int e = tid + 256 * i;
                int kl, col;
                if (FWD) { kl = e / (64 * NBW); col = e - kl * (64 * NBW); }
                else { kl = e & 31; col = e >> 5; }
                const bool ok = k0 + kl < Kc && col < Nc;
                Bs[kl * NCS + col] = ok ? vb[i] : 0.f;
                if (TAN) Us[kl * NCS + col] = ok ? -vu[i] : 0.f;
            }
            __syncthreads();
#pragma unroll 2
            for (int s = 0; s < GEN_KL / 4; ++s) {
                float av[4], rav[4];
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    av[rb] = As[(16 * rb + i16) * GEN_LA + 4 * s + kk];
                    if (TAN) rav[rb] = RAs[(16 * rb + i16) * GEN_LA + 4 * s + kk];
                }
#pragma unroll
                for (int c = 0; c < NBW; ++c) {
                    const int col = 16 * (w + 4 * c) + i16;
                    const float bv = Bs[(4 * s + kk) * NCS + col];
                    const float uv = TAN ? Us[(4 * s + kk) * NCS + col] : 0.f;
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) {
                        acc[rb][c] = mfma16(av[rb], bv, acc[rb][c]);
                        if (TAN) {
                            racc[rb][c] = mfma16(av[rb], uv, racc[rb][c]);
                            racc[rb][c] = mfma16(rav[rb], bv, racc[rb][c]);
                        }
                    }
                }
            }
        }
        if (FWD) {
            float* Hout = last ? a.mu : a.out_act[li + 1];
            float* RHout = last ? a.rmu : a.ract[li + 1];
#pragma unroll
            for (int c = 0; c < NBW; ++c) {
                const int col = 16 * (w + 4 * c) + i16;
                if (col < Nc) {
                    const float b = th[Ly.b_off + col];
                    const float ub = TAN ? -a.vdir[(long long)wk.task * a.NP + Ly.b_off + col] : 0.f;
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * rb + 4 * kk + r;
                            if (row < nrows) {
                                const float z = acc[rb][c][r] + b;
                                const int kind = last ? gen_out(a.act_kind) : gen_hidden(a.act_kind);
                                const float h = gen_act(kind, z);
                                Hout[(long long)(row0 + row) * Nc + col] = h;
                                if (TAN) {
                                    const float rz = racc[rb][c][r] + ub;
                                    RHout[(long long)(row0 + row) * Nc + col] = gen_act_d(kind, h) * rz;
                                }
                            }
                        }
                }
            }
        } else {
            const float* Hp = a.act[li];                 // this layer's input = the previous hidden layer's output
            const float* RHp = a.ract[li];
            float* DZo = a.dz[pp ^ 1];
            float* QZo = a.qz[pp ^ 1];
#pragma unroll
            for (int c = 0; c < NBW; ++c) {
                const int col = 16 * (w + 4 * c) + i16;
                if (col < Nc) {
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * rb + 4 * kk + r;
                            const long long o = (long long)(row0 + (row < nrows ? row : nrows - 1)) * Nc + col;
                            const float h = Hp[o], rh = TAN ? RHp[o] : 0.f, d1 = gen_act_d(gen_hidden(a.act_kind), h), dx = acc[rb][c][r];
                            if (row < nrows) {
                                DZo[o] = dx * d1;
                                if (TAN) QZo[o] = racc[rb][c][r] * d1 - (gen_hidden(a.act_kind) == GEN_ACT_TANH ? 2.f * dx * h * rh : 0.f);
                            }
                        }
                }
            }
        }
    }
}

// k_gen_wgrad: this work item's share of a layer's kernel / bias gradient into its partial row:
//   NT = 1 :  P[w_off + k N + n]  = sum_rows X[r][k] dZ[r][n]                 P[b_off + n] = sum_rows dZ[r][n]
//   NT = 2 :  ... = sum_rows R'X[r][k] dZ[r][n] + X[r][k] qZ[r][n]            ... = sum_rows qZ[r][n]
// The output is walked in slabs of 64 input units (4 blocks of 16) x the whole output width: 4 x NBW tiles per wave in
// registers; the rows are streamed through LDS once per slab (64-row chunks: the input slab [64][64] and the cotangents [64][N]).
// Sums run in row order inside a workgroup: bitwise reproducible.
// grid = work items, block = 256.
template <int NT, int NBW>
__global__ void __launch_bounds__(256) k_gen_wgrad(GenArgs a, int li, int pp) {
    PROMP_SMEM_DECL;
    float* Xs = (float*)PROMP_SMEM_PTR;                    // [NT][GEN_RW][GEN_LD]
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6), i16 = lane & 15, kk = lane >> 4;
    const WorkItem wk = a.work[blockIdx.x];
    const GenLin Ly = a.lin[li];
    const int K = Ly.K, N = Ly.N, NL = gen_wgrad_ld(N);
    float* Ds = Xs + NT * GEN_RW * GEN_LD;                  // [NT][GEN_RW][NL]
    float* P = a.partials + (long long)blockIdx.x * a.partial_stride;
    const float* X = a.act[li];
    const float* RX = (NT == 2 && li > 0) ? a.ract[li] : nullptr;
    const float* DZ = a.dz[pp];
    const float* QZ = NT == 2 ? a.qz[pp] : nullptr;
    // (one slab of 64 input units per blockIdx.y: the slabs' entries of the partial row are disjoint)
    for (int kb0 = GEN_KC * (int)blockIdx.y; kb0 < K; kb0 += GEN_KC * (int)gridDim.y) {
        f32x4 acc[4][NBW];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < NBW; ++c) acc[q][c] = zero4();
        float bsum = 0.f;                                  // bias gradient of column tid (first slab only)
        for (int row0 = wk.row_begin; row0 < wk.row_end; row0 += GEN_RW) {
            const int nrows = wk.row_end - row0 < GEN_RW ? wk.row_end - row0 : GEN_RW;
            // loads with clamped (always valid) addresses, all in flight together; zeros are selected on the way into LDS
            constexpr int NLX = GEN_RW * GEN_KC / 256;
            float vx[NLX], vrx[NLX];
#pragma unroll
            for (int i = 0; i < NLX; ++i) {
                const int e = tid + 256 * i, r = e >> 6, k = e & 63;
                const long long o = (long long)(row0 + (r < nrows ? r : nrows - 1)) * K + (kb0 + k < K ? kb0 + k : K - 1);
                vx[i] = X[o];
                if (NT == 2) vrx[i] = RX != nullptr ? RX[o] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NLX; ++i) {
                const int e = tid + 256 * i, r = e >> 6, k = e & 63;
                const bool ok = r < nrows && kb0 + k < K;
                Xs[r * GEN_LD + k] = ok ? vx[i] : 0.f;
                if (NT == 2) Xs[(GEN_RW + r) * GEN_LD + k] = ok ? vrx[i] : 0.f;
            }
#pragma unroll
            for (int cb = 0; cb < NBW; ++cb) {             // 64 output units at a time: [32][64] = 8 values per thread
                float vd[NLX], vq[NLX];
#pragma unroll
                for (int i = 0; i < NLX; ++i) {
                    const int e = tid + 256 * i, r = e >> 6, n = 64 * cb + (e & 63);
                    const long long o = (long long)(row0 + (r < nrows ? r : nrows - 1)) * N + (n < N ? n : N - 1);
                    vd[i] = DZ[o];
                    if (NT == 2) vq[i] = QZ[o];
                }
#pragma unroll
                for (int i = 0; i < NLX; ++i) {
                    const int e = tid + 256 * i, r = e >> 6, n = 64 * cb + (e & 63);
                    const bool ok = r < nrows && n < N;
                    Ds[r * NL + n] = ok ? vd[i] : 0.f;
                    if (NT == 2) Ds[(GEN_RW + r) * NL + n] = ok ? vq[i] : 0.f;
                }
            }
            __syncthreads();
            if (kb0 == 0 && tid < N) {
                const float* src = Ds + (NT == 2 ? GEN_RW * NL : 0) + tid;
                for (int r = 0; r < GEN_RW; ++r) bsum += src[r * NL];
            }
            // contraction over the chunk's rows: 16 steps of 4 rows; A[i = input unit][k = row], B[k = row][j = output unit]
            for (int s = 0; s < GEN_RW / 4; ++s) {
                const int r = 4 * s + kk;
                float av[4], rav[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    av[q] = Xs[r * GEN_LD + 16 * q + i16];
                    if (NT == 2) rav[q] = Xs[(GEN_RW + r) * GEN_LD + 16 * q + i16];
                }
#pragma unroll
                for (int c = 0; c < NBW; ++c) {
                    const int col = 16 * (w + 4 * c) + i16;
                    const float dv = col < N ? Ds[r * NL + col] : 0.f;
                    const float qv = (NT == 2 && col < N) ? Ds[(GEN_RW + r) * NL + col] : 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (NT == 1) acc[q][c] = mfma16(av[q], dv, acc[q][c]);
                        else {
                            acc[q][c] = mfma16(rav[q], dv, acc[q][c]);
                            acc[q][c] = mfma16(av[q], qv, acc[q][c]);
                        }
                    }
                }
            }
        }
        // D: col = i16 (output unit), row = 4 kk + r (input unit of block q)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < NBW; ++c) {
                const int col = 16 * (w + 4 * c) + i16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = kb0 + 16 * q + 4 * kk + r;
                    if (k < K && col < N) P[Ly.w_off + (long long)k * N + col] = acc[q][c][r];
                }
            }
        if (kb0 == 0 && tid < N) P[Ly.b_off + tid] = bsum;
    }
}

// Fixed-order sum of one float per thread over the workgroup (256 threads): lanes by xor shuffles, waves through LDS.
PROMP_DEV float gen_block_sum(float v, float* red, int tid) {
    v = wave_sum_f32(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// k_gen_loss: objective / KL of the work item's rows, the cotangents of the means (-> dz[pp], [rows][A]) and the log_std gradient;
// one thread per row, the row's actions in a loop.  HVP: also the tangents along u = -v (-> qz[pp]) and R'{d objective / d s},
// with kl_weight x the KL cotangents joined in (k_chain_hvp's loss level).  Writes P[oS .. oS + A), P[NP], P[NP + 1].
// What depends on the action only (the clipped log_std, its exponentials, the direction's entry; the old distribution's when it is
// per task) is tabulated in LDS once per workgroup.  The rows' log_std terms go through an LDS tile [256][A | 1], 256 rows at a
// time, and are summed per action by 4 x 64 threads (thread (q, j): rows q, q + 4, ... in row order; the four parts combined in
// fixed order at the end) -- until round 5 they went through global memory and A threads walked all rows of the work item one
// after the other: that tail was most of the kernel (46 us at 312 rows x 6 actions, 189 us at 625 x 17).
// grid = work items, block = 256, smem = gen_loss_smem(A).
#define GEN_LOSS_NC 8          // per-action constants
PROMP_HD size_t gen_loss_smem(int A) { return sizeof(float) * (size_t)(256 * (A | 1) + GEN_LOSS_NC * GEN_MAX_A + 4 * GEN_MAX_A); }
template <bool HVP, bool BWD>
__global__ void __launch_bounds__(256) k_gen_loss(GenArgs a, int pp) {
    PROMP_SMEM_DECL;
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const WorkItem wk = a.work[blockIdx.x];
    const int A = a.A, NP = a.NP, oS = NP - A, AS = A | 1;
    float* T = (float*)PROMP_SMEM_PTR;                 // [256][AS]
    float* cst = T + 256 * AS;                         // [GEN_LOSS_NC][GEN_MAX_A]: s, exp(-s), exp(2 s), mask, R's, so, exp(-so), exp(2 so)
    float* parts = cst + GEN_LOSS_NC * GEN_MAX_A;      // [4][GEN_MAX_A]
    const float* th = a.theta + (long long)wk.task * a.theta_task_stride;
    const int trow0 = a.task_row_offsets[wk.task], tn = a.task_row_offsets[wk.task + 1] - trow0;
    const float invN = 1.0f / (float)tn;
    float* P = a.partials + (long long)blockIdx.x * a.partial_stride;
    float loss = 0.f, klsum = 0.f;
    // (restrict-qualified copies: the row loops' loads may then be issued ahead of the stores of earlier actions)
    const float* __restrict__ g_mu = a.mu; const float* __restrict__ g_rmu = a.rmu; const float* __restrict__ g_act = a.actions;
    const float* __restrict__ g_om = a.old_mean;
    float* __restrict__ g_dz = a.dz[pp]; float* __restrict__ g_qz = a.qz[pp];
    if (tid < A) {
        const float sr = th[oS + tid];
        const bool clipped = a.clip_log_std && sr < a.min_log_std;
        const float s = clipped ? a.min_log_std : sr;
        cst[tid] = s;
        cst[GEN_MAX_A + tid] = expf(-s);
        cst[2 * GEN_MAX_A + tid] = expf(2.f * s);
        cst[3 * GEN_MAX_A + tid] = clipped ? 0.f : 1.f;
        cst[4 * GEN_MAX_A + tid] = (HVP && !clipped) ? -a.vdir[(long long)wk.task * NP + oS + tid] : 0.f;
        if (!a.ls_per_row) {
            const float so = a.old_log_std[(long long)wk.task * A + tid];
            cst[5 * GEN_MAX_A + tid] = so;
            cst[6 * GEN_MAX_A + tid] = expf(-so);
            cst[7 * GEN_MAX_A + tid] = expf(2.f * so);
        }
    }
    __syncthreads();
    float ssum = 0.f;          // sum of the log standard deviations (log-likelihood objective)
    for (int j = 0; j < A; ++j) ssum += cst[j];
    const bool is_kl = a.loss_kind == LOSS_KL, is_ratio = a.loss_kind == LOSS_RATIO, is_clip = a.loss_kind == LOSS_CLIP;
    const int out_kind = gen_out(a.act_kind);
    const int cj = tid & 63, cq = tid >> 6;
    float gacc = 0.f;          // this thread's share of the log_std gradient of action cj
    for (int base = wk.row_begin; base < wk.row_end; base += 256) {
        const bool rv = base + tid < wk.row_end;
        const int n = rv ? base + tid : wk.row_end - 1;            // (threads past the end shadow the last row and store nothing)
        const float* olsp = a.old_log_std + (long long)n * A;
        const float advn = a.adv[n];
        // pass 1 over the actions: the row's log-likelihood ratio, its tangent, the KL
        float dlp = 0.f, sumz2 = 0.f, kl = 0.f, Rlp = 0.f;
#pragma unroll 4
        for (int j = 0; j < A; ++j) {
            const float s = cst[j], e = cst[GEN_MAX_A + j], sn2 = cst[2 * GEN_MAX_A + j];
            const float mu = g_mu[(long long)n * A + j], ac = g_act[(long long)n * A + j], mo = g_om[(long long)n * A + j];
            float so, eo, so2;
            if (a.ls_per_row) { so = olsp[j]; eo = expf(-so); so2 = expf(2.f * so); }
            else { so = cst[5 * GEN_MAX_A + j]; eo = cst[6 * GEN_MAX_A + j]; so2 = cst[7 * GEN_MAX_A + j]; }
            const float z = (ac - mu) * e, zo = (ac - mo) * eo;
            const float num = (mo - mu) * (mo - mu) + so2 - sn2, rden = 1.0f / (2.f * sn2 + 1e-8f);
            dlp += (so - s) - 0.5f * (z * z - zo * zo);
            sumz2 += z * z;
            kl += num * rden + s - so;
            if (HVP) Rlp += z * e * g_rmu[(long long)n * A + j] + (z * z - 1.f) * cst[4 * GEN_MAX_A + j];
        }
        if (HVP && a.row_tan != nullptr && rv) a.row_tan[n] = Rlp;
        const float rho = expf(dlp), aw = advn * invN;
        const float x = rho * advn, y = fminf(fmaxf(rho, 1.f - a.clip_eps), 1.f + a.clip_eps) * advn;
        const float lp = -ssum - 0.5f * sumz2 - 0.5f * (float)A * 1.8378770664093453f;
        float c = -aw, lrow = -lp * aw;                                  // log-likelihood
        if (is_clip) { c = (x <= y) ? -aw * rho : 0.f; lrow = -fminf(x, y) * invN; }
        if (is_ratio) { c = -aw * rho; lrow = -rho * aw; }
        if (is_kl) { c = 0.f; lrow = kl * invN; }
        const float Rc = (HVP && is_ratio) ? c * Rlp : 0.f;
        loss += rv ? lrow : 0.f;
        klsum += rv ? kl * invN : 0.f;
        if (!BWD) continue;
        // pass 2: cotangents of the means (and their tangents), log_std terms
        __syncthreads();                 // the previous block's column sums are done with the tile
#pragma unroll 4
        for (int j = 0; j < A; ++j) {
            const float e = cst[GEN_MAX_A + j], sn2 = cst[2 * GEN_MAX_A + j], lmask = cst[3 * GEN_MAX_A + j];
            const float mu = g_mu[(long long)n * A + j], ac = g_act[(long long)n * A + j], mo = g_om[(long long)n * A + j];
            const float so2 = a.ls_per_row ? expf(2.f * olsp[j]) : cst[7 * GEN_MAX_A + j];
            const float z = (ac - mu) * e;
            const float num = (mo - mu) * (mo - mu) + so2 - sn2, den = 2.f * sn2 + 1e-8f, rden = 1.0f / den;
            const float dklm = -2.f * (mo - mu) * rden * invN;
            const float dkls = ((-2.f * sn2 * den - 4.f * num * sn2) * (rden * rden) + 1.f) * invN;
            float d, q = 0.f, os;
            if (!HVP) {
                d = c * z * e + (is_kl ? dklm : 0.f);
                os = c * (z * z - 1.f) + (is_kl ? dkls : 0.f);
            } else {
                const float Rs = cst[4 * GEN_MAX_A + j];
                const float Rmu = g_rmu[(long long)n * A + j];
                if (is_kl) {
                    // the objective is the mean KL itself (TRPO's constraint): see k_chain_hvp for the derivation
                    const float D = mo - mu, Pq = sn2 * (den + 2.f * num);
                    const float RP = 2.f * sn2 * Rs * (den + 2.f * num) - 4.f * sn2 * D * Rmu;
                    const float Rdm = 2.f * Rmu * rden + 8.f * D * sn2 * Rs * (rden * rden);
                    const float Rds = (-2.f * RP + 16.f * Pq * sn2 * Rs * rden) * (rden * rden);
                    d = dklm;
                    q = Rdm * invN;
                    os = Rds * invN;
                } else {
                    const float Rz = -Rmu * e - z * Rs;
                    const float Rd = Rc * z * e + c * (Rz * e - z * e * Rs);
                    const float Rds = Rc * (z * z - 1.f) + 2.f * c * z * Rz;
                    d = c * z * e;
                    q = Rd + a.kl_weight * dklm;
                    os = Rds + a.kl_weight * dkls;
                }
            }
            if (out_kind != GEN_ACT_IDENTITY) {      // output_nonlinearity (mlp.py:114-117): mu = f(z), d / dz = f'(z) d / dmu, and its R-operator
                const float d1 = gen_act_d(out_kind, mu);
                if (HVP) q = q * d1 - (out_kind == GEN_ACT_TANH ? 2.f * d * mu * g_rmu[(long long)n * A + j] : 0.f);
                d *= d1;
            }
            if (rv) {
                g_dz[(long long)n * A + j] = d;
                if (HVP) g_qz[(long long)n * A + j] = q;
            }
            T[tid * AS + j] = rv ? lmask * os : 0.f;
        }
        __syncthreads();
        if (cj < A)
            for (int r = cq; r < 256; r += 4) gacc += T[r * AS + cj];
    }
    const float L = gen_block_sum(loss, red, tid), Kl = gen_block_sum(klsum, red, tid);
    if (BWD) {
        // the log_std gradient: the four row parts of every action in fixed order
        if (cj < A) parts[cq * GEN_MAX_A + cj] = gacc;
        __syncthreads();
        if (tid < A) P[oS + tid] = (parts[tid] + parts[GEN_MAX_A + tid]) + (parts[2 * GEN_MAX_A + tid] + parts[3 * GEN_MAX_A + tid]);
    }
    if (tid == 0) {
        P[NP] = L;
        P[NP + 1] = Kl;
    }
}

// k_gen_policy_forward: k_policy_forward (promp_kernels_policy.h; MetaGaussianMLPPolicy.get_actions) for any layer table: the
// mean network of every task's current parameters on a small batch of observations, one thread per row, the hidden vectors
// ping-ponged through a scratch row pair in global memory (rollout-time inference: a few hundred rows per environment step).
// grid = tasks, block = 256
struct GenForwardArgs {
    const float* obs;          // [tasks][B][O]
    const float* theta_tasks;  // [tasks][Theta]
    float* mean;               // [tasks][B][A]
    float* scratch;            // [tasks][B][2][maxw]
    int B, NP, n_lin, maxw, act_kind;
    GenLin lin[GEN_MAX_LIN];
};

__global__ void __launch_bounds__(256) k_gen_policy_forward(GenForwardArgs a) {
    const int task = blockIdx.x;
    const float* th = a.theta_tasks + (long long)task * a.NP;
    for (int row = threadIdx.x; row < a.B; row += 256) {
        const long long r = (long long)task * a.B + row;
        const float* x = a.obs + r * a.lin[0].K;
        float* buf = a.scratch + r * 2 * a.maxw;
        for (int l = 0; l < a.n_lin; ++l) {
            const GenLin Ly = a.lin[l];
            const bool last = l == a.n_lin - 1;
            float* y = last ? a.mean + r * Ly.N : buf + (l & 1) * a.maxw;
            for (int j = 0; j < Ly.N; ++j) {
                float z = th[Ly.b_off + j];
                for (int k = 0; k < Ly.K; ++k) z = fmaf(x[k], th[Ly.w_off + k * Ly.N + j], z);
                y[j] = gen_act(last ? gen_out(a.act_kind) : gen_hidden(a.act_kind), z);
            }
            x = y;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Rollout-side kernels for any layer table (promp_kernels_rollout.h holds the two-layer forms and the
// contracts).  One WORKGROUP per environment: thread j owns unit j of a layer (kernel rows read coalesced, the layer's input
// broadcast from LDS), the layers follow one another behind a barrier -- an environment step at Humanoid's 376 x (64, 64) x 17 is
// ~500 dependent multiply-adds per thread instead of the 29 k a thread per environment would walk.  Every unit sums its inputs in
// index order with fmaf, as k_gen_policy_forward and mlp_mean do.
// ---------------------------------------------------------------------------------------------
PROMP_HD size_t gen_rollout_smem(int O) { return sizeof(float) * (size_t)(O + 2 * 256); }

// xs [lin[0].K] holds the observation; returns the mean [A] (in LDS, valid for every thread: the last barrier has been passed)
PROMP_DEV const float* gen_mlp_block(const float* th, const GenLin* lin, int n_lin, int act_kind, const float* xs, float* hs, int tid) {
    const float* in = xs;
    for (int l = 0; l < n_lin; ++l) {
        const GenLin Ly = lin[l];
        float* out = hs + (l & 1) * 256;
        if (tid < Ly.N) {
            float z = th[Ly.b_off + tid];
            for (int k = 0; k < Ly.K; ++k) z = fmaf(in[k], th[Ly.w_off + k * Ly.N + tid], z);
            out[tid] = gen_act(l == n_lin - 1 ? gen_out(act_kind) : gen_hidden(act_kind), z);
        }
        __syncthreads();
        in = out;
    }
    return in;
}

struct GenPolicyStepArgs {
    PolicyStepArgs p;          // k_policy_step's arguments (H1 / H2 unused)
    int n_lin, act_kind;
    GenLin lin[GEN_MAX_LIN];
};

// grid = (B, tasks), block = 256.  smem: gen_rollout_smem(O)
__global__ void __launch_bounds__(256) k_gen_policy_step(GenPolicyStepArgs g) {
    PROMP_SMEM_DECL;
    const PolicyStepArgs& a = g.p;
    float* xs = (float*)PROMP_SMEM_PTR;
    float* hs = xs + a.O;
    const int task = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
    const float* th = a.theta_tasks + (long long)task * a.NP;
    const int oS = a.NP - a.A;
    if (a.t == 0 && b == 0 && tid < a.A) {
        const float ls = th[oS + tid];
        a.old_ls[task * a.A + tid] = a.clip_infos ? fmaxf(ls, a.min_log_std) : ls;
    }
    const long long env = (long long)task * a.B + b, row = env * a.row_env_stride + a.t * a.row_t_stride;
    const float* x = a.obs_in + env * a.O;
    for (int k = tid; k < a.O; k += 256) {
        const float v = x[k];
        xs[k] = v;
        a.obs[row * a.O + k] = v;
    }
    __syncthreads();
    const float* mean = gen_mlp_block(th, g.lin, g.n_lin, g.act_kind, xs, hs, tid);
    const int j = 2 * tid;
    if (j < a.A) {
        float n0, n1;
        action_noise(a.seed, (unsigned long long)row, (unsigned)tid, a.stream, n0, n1);
        const float a0 = fmaf(expf(th[oS + j]), n0, mean[j]);
        a.mean[row * a.A + j] = mean[j];
        a.act[row * a.A + j] = a0;
        a.actions_out[env * a.A + j] = a0;
        if (j + 1 < a.A) {
            const float a1 = fmaf(expf(th[oS + j + 1]), n1, mean[j + 1]);
            a.mean[row * a.A + j + 1] = mean[j + 1];
            a.act[row * a.A + j + 1] = a1;
            a.actions_out[env * a.A + j + 1] = a1;
        }
    }
}

struct GenPointRolloutArgs {
    PointRolloutArgs p;        // k_point_rollout's arguments (H1 / H2 unused)
    int n_lin, act_kind;
    GenLin lin[GEN_MAX_LIN];
};

// grid = (B, tasks), block = 256: the environment's state lives in thread 0, the network runs on the workgroup.
// smem: gen_rollout_smem(2)
__global__ void __launch_bounds__(256) k_gen_point_rollout(GenPointRolloutArgs g) {
    PROMP_SMEM_DECL;
    const PointRolloutArgs& a = g.p;
    float* xs = (float*)PROMP_SMEM_PTR;
    float* hs = xs + 2;
    const int task = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
    const float* th = a.theta_tasks + (long long)task * a.NP;
    const int oS = a.NP - 2;
    const float ls0 = th[oS], ls1 = th[oS + 1];
    if (b == 0 && tid == 0) {
        a.old_ls[task * 2 + 0] = a.clip_infos ? fmaxf(ls0, a.min_log_std) : ls0;
        a.old_ls[task * 2 + 1] = a.clip_infos ? fmaxf(ls1, a.min_log_std) : ls1;
    }
    const float sd0 = expf(ls0), sd1 = expf(ls1);
    const double g0 = a.goals[task * 2], g1 = a.goals[task * 2 + 1];
    const long long env = (long long)task * a.B + b;
    double s0 = a.start[env * 2], s1 = a.start[env * 2 + 1];       // (advanced by thread 0 only)
    for (int t = 0; t < a.T; ++t) {
        const long long row = env * a.T + t;
        if (tid == 0) {
            xs[0] = (float)s0;
            xs[1] = (float)s1;
        }
        __syncthreads();
        const float* m = gen_mlp_block(th, g.lin, g.n_lin, g.act_kind, xs, hs, tid);
        if (tid == 0) {
            float n0, n1;
            if (a.noise != nullptr) {
                n0 = a.noise[row * 2];
                n1 = a.noise[row * 2 + 1];
            } else {
                action_noise(a.seed, (unsigned long long)row, 0u, a.stream, n0, n1);
            }
            const float a0 = fmaf(sd0, n0, m[0]), a1 = fmaf(sd1, n1, m[1]);
            a.obs[row * 2] = xs[0];  a.obs[row * 2 + 1] = xs[1];
            a.mean[row * 2] = m[0];  a.mean[row * 2 + 1] = m[1];
            a.act[row * 2] = a0;  a.act[row * 2 + 1] = a1;
            const double r = point_env_step(a, s0, s1, g0, g1, a0, a1);
            a.rew[row] = (float)r;
        }
        __syncthreads();       // the next step's observation and hidden vectors overwrite what thread 0 has just read
    }
}
