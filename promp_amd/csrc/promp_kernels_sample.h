// promp_kernels_sample.h -- sample processing on the device (reference rows a1-a7), FP64 like the
// reference (SciPy lfilter / NumPy lstsq compute in float64).
//
//   k_returns   : per path  R[t] = r[t] + g R[t+1]             utils/utils.py:74-81, samplers/base.py:103-104
//   k_gram      : per task  [Phi R]^T [Phi R] (normal equations) baselines/linear_baseline.py:66-67,101-106
//   k_fit       : per task  (G + reg I) w = Phi^T R, NaN retry   baselines/linear_baseline.py:68-77
//   k_gae       : per path  b = Phi w ; delta ; GAE scan          baselines/linear_baseline.py:17-33, samplers/base.py:151-162
//   k_normalize : per task  (adv-mean)/(std+1e-8) [, shift >0]    utils/utils.py:59-71
//
// One wavefront owns one path in the scans: the linear recurrence y[t] = x[t] + c*y[t+1] is a
// 6-step wave-level suffix scan over 64 time steps (weights c, c^2, c^4, ...), chained across
// 64-step chunks from the end of the path with a carry, so ragged path lengths need no padding.
// The Gram matrix runs on the FP64 matrix cores (v_mfma_f64_16x16x4_f64): the target is appended
// to the features as one more column so that Phi^T R falls out of the same product.
#pragma once
#include <utility>
#include "promp_device.h"
#include "promp_kernels_policy.h"  // WorkItem

enum { BASE_ZERO = 0, BASE_LINFEAT = 1, BASE_LINTIME = 2 };

struct SampleArgs {
    const float* obs;              // [rows][O]
    const float* rew;              // [rows]
    const double* rew64;           // optional [rows]: the rewards in the environment's float64 (else NULL: rew is used)
    const int* path_row_offsets;   // [paths+1]
    const int* path_task;          // [paths]
    const int* row_t;              // [rows] time index inside the path
    const int* task_row_offsets;   // [tasks+1]
    const int* task_path_offsets;  // [tasks+1]
    const WorkItem* work;          // row ranges (multiples of 64 from the task start)
    const int* task_wg_offsets;    // [tasks+1]
    int O, D, kind;
    double gamma, lam, reg;
    int normalize, positive;
    double* ret64;
    float* ret32;
    double* adv64;
    float* adv32;
    double* path_ret0;
    double* path_undisc;
    double* path_rsq;
    double* path_mom;       // [paths][3] sum adv, sum adv^2, min adv
    double* gram_partials;  // [grid][NPAIR*256]
    double* coeffs;         // [tasks][coeff_stride]
    int coeff_stride;
    double* bl64;           // optional [rows]: the baseline predictions Phi.w (LinearBaseline.predict), else NULL
};

// grid = paths, block = 64
__global__ void __launch_bounds__(64) k_returns(SampleArgs a) {
    const int p = blockIdx.x, lane = threadIdx.x;
    const int row0 = a.path_row_offsets[p], T = a.path_row_offsets[p + 1] - row0;
    const double g = a.gamma;
    const double f = pow(g, (double)(64 - lane));
    double carry = 0.0, usum = 0.0, rsq = 0.0, y0 = 0.0;
    for (int end = T; end > 0; end -= 64) {
        const int t = end - 64 + lane;
        const int rr = row0 + (t >= 0 ? t : 0);
        const double x = (t >= 0) ? (a.rew64 ? a.rew64[rr] : (double)a.rew[rr]) : 0.0;
        double y = x, gg = g;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double up = shfl_down_f64(y, d);
            if (lane + d < 64) y += gg * up;
            gg *= gg;
        }
        y += f * carry;
        if (t >= 0) {
            a.ret64[row0 + t] = y;
            a.ret32[row0 + t] = (float)y;
        }
        if (t == 0) y0 = y;
        carry = shfl_idx_f64(y, 0);
        usum += x;
        rsq += x * x;
    }
    usum = wave_sum_f64(usum);
    rsq = wave_sum_f64(rsq);
    y0 = wave_sum_f64(y0);
    if (lane == 0) {
        a.path_ret0[p] = y0;
        a.path_undisc[p] = usum;
        a.path_rsq[p] = rsq;
    }
}

// DiCE weights (meta_algos/dice_maml.py:39-45, 245-258: the magic box couples the time steps of a path).
//   mode 0:  out[t] = sum_{t' >= t} rw[t']                                   (gradient weights w)
//   mode 1:  out[t] = sum_{t' >= t} rw[t'] C[t'],  C[t'] = sum_{t'' <= t'} c[t'']   (Hessian-vector coupling weights u)
// One wave per path, float64 suffix scans in 64-row chunks from the end (as k_returns with discount 1); the prefix sum
// is total - suffix + own.  grid = paths, block = 64.
struct DiceScanArgs {
    const int* path_row_offsets;
    const float* rw;       // [rows] adjusted reward * rows / (paths * max_path_length) of every valid row
    const float* c;        // [rows] row tangents (mode 1)
    float* out;            // [rows]
    double* tmp;           // [rows] scratch (mode 1)
    int mode;
};
PROMP_DEV double dice_suffix_chunk(double x, int lane) {
    double y = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double up = shfl_down_f64(y, d);
        if (lane + d < 64) y += up;
    }
    return y;
}
__global__ void __launch_bounds__(64) k_dice_scan(DiceScanArgs a) {
    const int p = blockIdx.x, lane = threadIdx.x;
    const int row0 = a.path_row_offsets[p], T = a.path_row_offsets[p + 1] - row0;
    double total = 0.0;
    if (a.mode == 1) {
        double carry = 0.0;
        for (int end = T; end > 0; end -= 64) {
            const int t = end - 64 + lane;
            const double x = (t >= 0) ? (double)a.c[row0 + t] : 0.0;
            const double y = dice_suffix_chunk(x, lane) + carry;
            if (t >= 0) a.tmp[row0 + t] = y;
            carry = shfl_idx_f64(y, 0);
        }
        total = carry;
    }
    double carry = 0.0;
    for (int end = T; end > 0; end -= 64) {
        const int t = end - 64 + lane;
        double x = 0.0;
        if (t >= 0) {
            x = (double)a.rw[row0 + t];
            if (a.mode == 1) x *= total - a.tmp[row0 + t] + (double)a.c[row0 + t];     // C[t]
        }
        const double y = dice_suffix_chunk(x, lane) + carry;
        if (t >= 0) a.out[row0 + t] = (float)y;
        carry = shfl_idx_f64(y, 0);
    }
}

// k_gram: partial Gram of [Phi R] over a work item's rows, wave-private like the policy passes: every wave builds
// the features of its own 16-row chunks in its own LDS tile and feeds them to v_mfma_f64_16x16x4_f64 (the same
// register is A and B operand of a block pair); no workgroup barrier in the chunk loop.  The waves' blocks are added
// in wave order through LDS slabs at the end.
template <int NBLK>
struct GramCfg {
    static constexpr int NW = (NBLK <= 3) ? 8 : 4;                            // waves per workgroup
    static constexpr int FS = (NBLK % 2 == 1) ? 16 * NBLK : 16 * NBLK + 16;   // feature-tile row stride (doubles)
    static constexpr int NPAIR = NBLK * (NBLK + 1) / 2;
    static constexpr int WAVE_BYTES = 16 * FS * 8 + 16 * 32 * 4 + 2 * 16 * 8;  // Phi | raw obs [16][<=32] | targets | tau
    static constexpr int SLAB_BYTES = NW * NPAIR * 256 * 8;
    static constexpr int SMEM_BYTES = (NW * WAVE_BYTES > SLAB_BYTES) ? NW * WAVE_BYTES : SLAB_BYTES;
};

// grid = work items, block = 64 * GramCfg<NBLK>::NW
#ifdef PROMP_DEV_STAMPS
#define SA_STAMPS unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tph = promp_clock()
#define SA_PHASE(i) do { const unsigned long long now_ = promp_clock(); ph[i] += now_ - tph; tph = now_; } while (0)
#else
#define SA_STAMPS do { } while (0)
#define SA_PHASE(i) do { } while (0)
#endif
template <int NBLK>
__global__ void __launch_bounds__(64 * GramCfg<NBLK>::NW) k_gram(SampleArgs a) {
    SA_STAMPS;
    constexpr int NW = GramCfg<NBLK>::NW, FS = GramCfg<NBLK>::FS, NPAIR = GramCfg<NBLK>::NPAIR, NC = 16 * NBLK;
    PROMP_SMEM_DECL;
    unsigned char* smem = (unsigned char*)PROMP_SMEM_PTR;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i16 = lane & 15, kk = lane >> 4;
    double* Phi = (double*)(smem + (size_t)w * GramCfg<NBLK>::WAVE_BYTES);
    const WorkItem wk = a.work[blockIdx.x];
    const int O = a.O, D = a.D;
    f64x4 acc[NPAIR];
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) acc[p] = zero4d();
    // A chunk's inputs (this lane's share of the [16][O <= 32] observation rows, a row's return and time index) are
    // requested one chunk ahead: the wave walks ~5 chunks and would otherwise sit out a memory round trip in each.
    float xr[8];
    double tr = 0.0;
    int ttr = 0;
    auto request = [&](int base) {
        const bool live = base < wk.row_end;
        const int nrows = !live ? 0 : (wk.row_end - base) < 16 ? (wk.row_end - base) : 16;
        const long long b0 = live ? base : wk.row_begin;          // always a valid row of this work item
        if (a.kind == BASE_LINFEAT) {
            const int lim = nrows * O;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = lane + 64 * u;
                const float x = a.obs[b0 * O + (e < lim ? e : 0)];
                xr[u] = (e < lim) ? x : 0.f;
            }
        }
        if (lane < 16) {
            const int r = lane < nrows ? lane : 0;
            tr = a.ret64[b0 + r];
            ttr = a.row_t[b0 + r];
        }
    };
    // The feature tile is written straight from the registers the request filled: element e = lane + 64 u of the [16][O] chunk is
    // observation c = e mod O of row e / O, and goes to columns c (clipped) and O + c (its square) -- the positions are the same for
    // every chunk.  Lanes 0..15 add their row's time features, the constant and the target.  Columns past D + 1 stay zero.
    int poff[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = lane + 64 * u, r = e / (O > 0 ? O : 1);
        poff[u] = (a.kind == BASE_LINFEAT && e < 16 * O) ? r * FS + (e - r * O) : -1;
    }
    for (int e = lane; e < 16 * FS; e += 64) Phi[e] = 0.0;
    wave_sync();
    const int qt = (a.kind == BASE_LINFEAT) ? 2 * O : 0;        // first of the four time columns (tau, tau^2, tau^3, 1)
    request(wk.row_begin + 16 * w);
    SA_PHASE(0);
    for (int base = wk.row_begin + 16 * w; base < wk.row_end; base += 16 * NW) {
        const int nrows = (wk.row_end - base) < 16 ? (wk.row_end - base) : 16;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (poff[u] >= 0) {
                // (rows past the chunk's last arrive as zeros.)  the reference squares in the observations' own dtype (float32),
                // then promotes
                const float oc = fminf(fmaxf(xr[u], -10.f), 10.f);
                Phi[poff[u]] = (double)oc;
                Phi[poff[u] + O] = (double)(oc * oc);
            }
        if (lane < 16) {
            const bool rv = lane < nrows;
            const double tau = rv ? (double)ttr / 100.0 : 0.0;
            double* pr = Phi + lane * FS;
            pr[qt] = tau;
            pr[qt + 1] = tau * tau;
            pr[qt + 2] = tau * tau * tau;
            pr[qt + 3] = rv ? 1.0 : 0.0;
            pr[D] = rv ? tr : 0.0;
        }
        SA_PHASE(1);
        request(base + 16 * NW);
        wave_sync();
        SA_PHASE(2);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            double av[NBLK];
#pragma unroll
            for (int b = 0; b < NBLK; ++b) av[b] = Phi[(4 * s + kk) * FS + 16 * b + i16];
            int p = 0;
#pragma unroll
            for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
                for (int bj = bi; bj < NBLK; ++bj) {
                    acc[p] = mfma16d(av[bi], av[bj], acc[p]);
                    ++p;
                }
        }
        wave_sync();
        SA_PHASE(3);
    }
    __syncthreads();
    SA_PHASE(4);
    double* S = (double*)smem;   // [NW][NPAIR*256]
#pragma unroll
    for (int p = 0; p < NPAIR; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(w * NPAIR + p) * 256 + (kk + 4 * r) * 16 + i16] = acc[p][r];
    __syncthreads();
    double* out = a.gram_partials + (long long)blockIdx.x * (NPAIR * 256);
    for (int e = tid; e < NPAIR * 256; e += 64 * NW) {
        double t = 0.0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) t += S[ww * NPAIR * 256 + e];
        out[e] = t;
    }
    SA_PHASE(5);
#ifdef PROMP_DEV_STAMPS
    if (blockIdx.x == 0 && tid == 0)
        printf("k_gram cycles (wave 0 of workgroup 0): entry %llu | features %llu | request + sync %llu | products %llu | barrier %llu | slab sums %llu\n",
               ph[0], ph[1], ph[2], ph[3], ph[4], ph[5]);
#endif
}

// grid = tasks, block = 256.  smem: G[(D+1)^2] + Wm[(D+1)^2] + yv[D+1] + wv[D+1] + dg[D+1]  (doubles)
// The factorisation is a chain of D dependent column steps; per step: every thread recomputes the pivot (no
// broadcast barrier), rows scale their column entry, ONE barrier, the trailing update spread over 256 threads,
// ONE barrier.  The diagonal goes to a side array so nobody reads a value another thread is replacing.
__global__ void __launch_bounds__(256) k_fit(SampleArgs a, int NBLK) {
    PROMP_SMEM_DECL;
    const int D = a.D, DA = D + 1;
    double* G = (double*)PROMP_SMEM_PTR;
    double* Wm = G + DA * DA;
    double* yv = Wm + DA * DA;
    double* wv = yv + DA;
    double* dg = wv + DA;
    int* flag = (int*)(dg + DA);
    const int tid = threadIdx.x, task = blockIdx.x;
    const int NPAIR = NBLK * (NBLK + 1) / 2;
    // 1. sum the task's partial Gram blocks in workgroup order and scatter into the symmetric matrix
    const int wg0 = a.task_wg_offsets[task], wg1 = a.task_wg_offsets[task + 1];
    for (int e = tid; e < NPAIR * 256; e += 256) {
        double s = 0.0;
#pragma unroll 4
        for (int wg = wg0; wg < wg1; ++wg) s += a.gram_partials[(long long)wg * (NPAIR * 256) + e];
        int p = e >> 8, bi = 0, rem = p;
        while (rem >= NBLK - bi) {
            rem -= NBLK - bi;
            ++bi;
        }
        const int bj = bi + rem;
        const int row = 16 * bi + ((e & 255) >> 4), col = 16 * bj + (e & 15);
        if (row < DA && col < DA) {
            G[row * DA + col] = s;
            if (bi != bj) G[col * DA + row] = s;
        }
    }
    __syncthreads();
    // 2. Cholesky of (G[:D,:D] + reg I) carrying the right-hand-side row D along (forward solve for free),
    //    then back substitution; NaN => reg *= 10, at most 5 tries (linear_baseline.py:68-77)
    const float rDA = 1.0f / (float)DA;
    double reg = a.reg;
    for (int attempt = 0; attempt < 5; ++attempt) {
        for (int e = tid; e < DA * DA; e += 256) {
            const int i = (int)(((float)e + 0.5f) * rDA), j = e - i * DA;
            Wm[e] = G[e] + ((i == j && i < D) ? reg : 0.0);
        }
        __syncthreads();
        for (int j = 0; j < D; ++j) {
            const double piv = sqrt(Wm[j * DA + j]);
            if (tid > j && tid <= D) Wm[tid * DA + j] /= piv;
            if (tid == j) dg[j] = piv;
            __syncthreads();
            const int nr = D - j, nc = D - j - 1;  // rows j+1..D, cols j+1..D-1
            const float rnc = 1.0f / (float)(nc > 0 ? nc : 1);
            for (int e = tid; e < nr * nc; e += 256) {
                const int io = (int)(((float)e + 0.5f) * rnc);
                const int i = j + 1 + io, k = j + 1 + (e - io * nc);
                if (k <= i) Wm[i * DA + k] -= Wm[i * DA + j] * Wm[k * DA + j];
            }
            __syncthreads();
        }
        if (tid < D) yv[tid] = Wm[D * DA + tid];
        __syncthreads();
        for (int j = D - 1; j >= 0; --j) {
            const double wj = yv[j] / dg[j];
            if (tid == 0) wv[j] = wj;
            if (tid < j) yv[tid] -= Wm[j * DA + tid] * wj;
            __syncthreads();
        }
        if (tid == 0) *flag = 0;
        __syncthreads();
        if (tid < D && wv[tid] != wv[tid]) *flag = 1;
        __syncthreads();
        const int bad = *flag;
        __syncthreads();
        if (!bad) break;
        reg *= 10.0;
    }
    if (tid < D) a.coeffs[(long long)task * a.coeff_stride + tid] = wv[tid];
}

// One column step of k_fit_wave's factorisation (J is a compile-time constant: the steps are instantiated one by one, which
// keeps the row in registers whatever the unroller's size limits say).
#define FITWV_CS 65        // doubles between the factor's columns in LDS (k_fit_wave: 64 lanes + 1: a lane walking down its own
                           // column -- the back substitution's operands -- is then on its own pair of banks)
PROMP_HD size_t fitwv_aux(int dt) { return (size_t)dt * FITWV_CS + 2 + 64; }     // the factor by columns + 1 / L[j][j]
template <int J, int DT>
PROMP_DEV void fitw_column(double (&W)[DT], double* colbuf, double* rsv, double& rs, int lane, int D) {
    if (J >= D) return;       // (wave-uniform)
    // pivot and its reciprocal from ONE reciprocal square root (sqrt followed by a division is two long software sequences, this
    // is one); `rs` arrives computed: the previous step started it as soon as column J was final (see below).  Lane J holds the
    // pivot element d, so one multiplication makes its sqrt(d) = d rs there and L[k][J] in the lanes below: no select.
    W[J] *= rs;
    // The finished column also goes to LDS: the columns together are the factor the back substitution reads (no transposition pass).
    colbuf[J * FITWV_CS + lane] = W[J];
    rsv[J] = rs;              // (every lane, the same value: the back substitution reads lane j's reciprocal from here)
    // lane k holds L[k][J] in W[J].  The first trailing column goes alone: it completes column J + 1, whose pivot's reciprocal
    // square root (seed + one correction: five dependent float64 operations) is then issued in pieces BETWEEN the batches of this
    // step's remaining updates (a wave issues in order: the pieces only overlap with independent work that sits between them in
    // the instruction stream; sched_fence pins that order).
    double dn = 1.0, r0 = 1.0, e = 0.0;
    if (J + 1 < DT) {
        W[J + 1] -= W[J] * readlane_f64(W[J], J + 1);
        dn = readlane_f64(W[J + 1], J + 1);
        r0 = rsq_seed(dn);
    }
    // Eight multipliers are read into eight scalar pairs before the first product: a plain loop compiles to "readlane, readlane,
    // s_nop, fma" on ONE pair (41 cycles per column).  What is left is the price of the multipliers themselves: two v_readlane
    // and the product cost 30 cycles per trailing column in any order (990 columns: the 29 k cycles of this phase, measured).
    // Broadcast reads of the LDS copy instead (ds_read2_b64 off one opaque address register, four in flight) were tried and are
    // slower: the LDS round trip sits on every step's dependent path, 34.5 k cycles.
    constexpr int NBATCH = (DT - (J + 2) + 7) / 8;
#pragma unroll
    for (int b = 0; b < (NBATCH > 0 ? NBATCH : 0); ++b) {
        const int k0 = J + 2 + 8 * b;
        double m[8];
        sched_fence();
#pragma unroll
        for (int u = 0; u < 8; ++u) m[u] = (k0 + u < DT) ? readlane_f64(W[J], k0 + u) : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < DT) W[k0 + u] -= W[J] * m[u];
        sched_fence();
        if (b == 0) e = rsq_e(dn, r0);
        if (b == 1 || (b == 0 && NBATCH == 1)) rs = rsq_finish(r0, e);
    }
    if (NBATCH <= 0) rs = rsq_finish(r0, rsq_e(dn, r0));
}
template <int DT, int... Js>
PROMP_DEV void fitw_columns(double (&W)[DT], double* colbuf, double* rsv, int lane, int D, std::integer_sequence<int, Js...>) {
    const double d0 = readlane_f64(W[0], 0), s0 = rsq_seed(d0);
    double rs = rsq_finish(s0, rsq_e(d0, s0));
    (fitw_column<Js, DT>(W, colbuf, rsv, rs, lane, D), ...);
}

// k_fit_wave: the same solve as k_fit, one WAVE per task, for D + 1 <= 64 (obs_dim <= 29).
//
// Lane i holds row i of the work matrix in registers (row D = the right-hand side, carried along as in k_fit).  A column
// step of the Cholesky needs the pivot and, per trailing column k, the freshly scaled entry L[k][j] as a wave-uniform
// multiplier: both come out of the owning lane with v_readlane (scalar registers), so a step is
// "readlane, readlane, v_fma_f64" per trailing column with no barrier and no LDS round trip on the dependent chain
// (k_fit: two workgroup barriers plus LDS traffic per column, 49 us at 45 x 45; this: ~12 us).  All lanes update all
// trailing columns (the upper triangle is never read), so there is no per-lane masking either.  The back substitution
// runs over the transposed factor: the rows are written to LDS once and read back by column.
// Same elimination order as k_fit; the column scaling multiplies by 1/sqrt(pivot) instead of dividing by sqrt(pivot), so
// the two kernels agree to rounding (1e-15 relative), not bit for bit.
// DT = compile-time bound on D + 1 (the j / k loops are fully unrolled over it; steps j >= D leave by a uniform branch).
// grid = tasks, block = FITWV_NT: all waves sum the task's partial Gram blocks (many loads in flight), wave 0 factorizes.
#define FITWV_NT 512       // threads of k_fit_wave: eight waves sum the partial blocks, wave 0 factorizes
template <int DT>
__global__ void __launch_bounds__(FITWV_NT) k_fit_wave(SampleArgs a, int NBLK) {
    PROMP_SMEM_DECL;
    SA_STAMPS;
    const int D = a.D, DA = D + 1;
    double* colbuf = (double*)PROMP_SMEM_PTR;     // [DT][FITWV_CS] the factor by columns: column j = lanes' L[lane][j] (fitw_column)
    double* rsv = colbuf + DT * FITWV_CS + 2;     // [64] 1 / L[j][j]
    double* G = colbuf + fitwv_aux(DT);           // [DA][DA] symmetric Gram matrix (kept for the retries)
    const int lane = threadIdx.x & 63, task = blockIdx.x;
    const int NPAIR = NBLK * (NBLK + 1) / 2;
    // 1. sum the task's partial Gram blocks in workgroup order and scatter into the symmetric matrix
    const int wg0 = a.task_wg_offsets[task], wg1 = a.task_wg_offsets[task + 1];
    // Every entry a thread owns (up to five: DA <= 64 means at most ten block pairs) and sixteen partial blocks of each are
    // requested before the first is added (indices past the task's last workgroup read that workgroup again and add nothing):
    // the partial blocks were written by other XCDs a launch ago, a round trip costs ~3 k cycles, and a task has ~14 of them.
    {
        constexpr int NE = 5;
        double sacc[NE];
#pragma unroll
        for (int q = 0; q < NE; ++q) sacc[q] = 0.0;
        for (int base = wg0; base < wg1; base += 16) {
            double v[NE][16];
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                const int e = threadIdx.x + q * FITWV_NT;
                if (e < NPAIR * 256) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int wg = base + u < wg1 ? base + u : wg1 - 1;
                        v[q][u] = a.gram_partials[(long long)wg * (NPAIR * 256) + e];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                const int e = threadIdx.x + q * FITWV_NT;
                if (e < NPAIR * 256) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) sacc[q] += (base + u < wg1) ? v[q][u] : 0.0;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int e = threadIdx.x + q * FITWV_NT;
            if (e < NPAIR * 256) {
                int p = e >> 8, bi = 0, rem = p;
                while (rem >= NBLK - bi) {
                    rem -= NBLK - bi;
                    ++bi;
                }
                const int bj = bi + rem;
                const int row = 16 * bi + ((e & 255) >> 4), col = 16 * bj + (e & 15);
                if (row < DA && col < DA) {
                    G[row * DA + col] = sacc[q];
                    if (bi != bj) G[col * DA + row] = sacc[q];
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    SA_PHASE(0);
    const int row = lane < DA ? lane : DA - 1;    // (lanes >= DA shadow the last row: finite values, never read back)
    double reg = a.reg;
    double wsol = 0.0;
    for (int attempt = 0; attempt < 5; ++attempt) {
        double W[DT];
#pragma unroll
        for (int k = 0; k < DT; ++k) W[k] = (k < D) ? G[row * DA + k] + ((k == row) ? reg : 0.0) : 0.0;
        // 2. Cholesky of (G[:D,:D] + reg I) carrying the right-hand-side row D along (forward solve for free)
        fitw_columns<DT>(W, colbuf, rsv, lane, D, std::make_integer_sequence<int, DT>());
        SA_PHASE(1);
        // 3. back substitution over the transposed factor: y[t] -= L[j][t] w[j], j = D-1 .. 0 (lane t holds y[t]).  Lane t's
        // operands L[j][t], j > t, are column t of the factor -- its own slot of the column store, read back into the registers the
        // factorisation has just freed (zeros for j <= t: y[t] then stays what it was when step t used it).  The dependent steps
        // contain no memory access, no select and no division: w[j] = y[j] / L[j][j] is a multiplication by the stored reciprocal.
        wave_fence();
        double y = (lane < D) ? colbuf[lane * FITWV_CS + D] : 0.0;
        const double rdg = rsv[lane < D ? lane : 0];        // lane j: 1 / L[j][j] (the reciprocal square root of its column step)
#pragma unroll
        for (int k = 0; k < DT; ++k) {
            const double l = colbuf[row * FITWV_CS + (k < DA ? k : 0)];
            W[k] = (k < D && row < k) ? l : 0.0;
        }
        SA_PHASE(2);
#pragma unroll
        for (int j = DT - 1; j >= 0; --j) {
            if (j < D) {
                const double wj = readlane_f64(y * rdg, j);
                y -= W[j] * wj;
            }
        }
        wsol = y * rdg;
        wave_sync();
        SA_PHASE(3);
        if (!wave_any(lane < D && wsol != wsol)) break;      // NaN => reg *= 10, at most 5 tries (linear_baseline.py:68-77)
        reg *= 10.0;
    }
    if (lane < D) a.coeffs[(long long)task * a.coeff_stride + lane] = wsol;
#ifdef PROMP_DEV_STAMPS
    if (task == 0 && lane == 0)
        printf("k_fit_wave cycles: partial sums %llu | factorisation %llu | column reads %llu | back substitution %llu\n", ph[0], ph[1], ph[2], ph[3]);
#endif
}

// grid = paths, block = 64.  smem: D doubles (the task's coefficients)
__global__ void __launch_bounds__(64) k_gae(SampleArgs a) {
    PROMP_SMEM_DECL;
    double* wc = (double*)PROMP_SMEM_PTR;
    const int p = blockIdx.x, lane = threadIdx.x;
    const int row0 = a.path_row_offsets[p], T = a.path_row_offsets[p + 1] - row0;
    const int task = a.path_task[p];
    const int O = a.O, D = a.D;
    if (a.kind != BASE_ZERO)
        for (int i = lane; i < D; i += 64) wc[i] = a.coeffs[(long long)task * a.coeff_stride + i];
    __syncthreads();
    const double g = a.gamma, gl = a.gamma * a.lam;
    const double f = pow(gl, (double)(64 - lane));
    double carry = 0.0, bcarry = 0.0, s1 = 0.0, s2 = 0.0, mn = 1e300;
    for (int end = T; end > 0; end -= 64) {
        const int t = end - 64 + lane;
        const bool valid = t >= 0;
        const long long row = (long long)row0 + t;
        double b = 0.0;
        if (valid && a.kind != BASE_ZERO) {
            int q = 0;
            if (a.kind == BASE_LINFEAT) {
                for (int c = 0; c < O; ++c) {
                    const float o = a.obs[row * O + c];
                    const float oc = fminf(fmaxf(o, -10.f), 10.f);
                    b += wc[c] * (double)oc + wc[O + c] * (double)(oc * oc);
                }
                q = 2 * O;
            }
            const double tau = (double)t / 100.0;
            b += wc[q] * tau + wc[q + 1] * (tau * tau) + wc[q + 2] * (tau * tau * tau) + wc[q + 3];
        }
        double bn = shfl_down_f64(b, 1);
        if (lane == 63) bn = bcarry;
        const double rw = a.rew64 ? a.rew64[valid ? row : 0] : (double)a.rew[valid ? row : 0];
        const double x = valid ? (rw + g * bn - b) : 0.0;
        double y = x, gg = gl;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double up = shfl_down_f64(y, d);
            if (lane + d < 64) y += gg * up;
            gg *= gg;
        }
        y += f * carry;
        if (valid) {
            if (a.bl64 != nullptr) a.bl64[row] = b;
            a.adv64[row] = y;
            s1 += y;
            s2 += y * y;
            mn = y < mn ? y : mn;
        }
        carry = shfl_idx_f64(y, 0);
        bcarry = shfl_idx_f64(b, 0);
    }
    s1 = wave_sum_f64(s1);
    s2 = wave_sum_f64(s2);
    mn = wave_min_f64(mn);
    if (lane == 0) {
        a.path_mom[3 * p + 0] = s1;
        a.path_mom[3 * p + 1] = s2;
        a.path_mom[3 * p + 2] = mn;
    }
}

// grid = work items, block = 256
__global__ void __launch_bounds__(256) k_normalize(SampleArgs a) {
    __shared__ double st[3];
    const int tid = threadIdx.x;
    const WorkItem wk = a.work[blockIdx.x];
    const int task = wk.task;
    if (tid == 0) {
        double s1 = 0.0, s2 = 0.0, mn = 1e300;
        for (int p = a.task_path_offsets[task]; p < a.task_path_offsets[task + 1]; ++p) {
            s1 += a.path_mom[3 * p];
            s2 += a.path_mom[3 * p + 1];
            mn = a.path_mom[3 * p + 2] < mn ? a.path_mom[3 * p + 2] : mn;
        }
        const double n = (double)(a.task_row_offsets[task + 1] - a.task_row_offsets[task]);
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        st[0] = mean;
        st[1] = sqrt(var);
        st[2] = mn;
    }
    __syncthreads();
    const double mean = st[0], sd = st[1];
    double mn = st[2];
    if (a.normalize) mn = (mn - mean) / (sd + 1e-8);
    for (int row = wk.row_begin + tid; row < wk.row_end; row += 256) {
        double x = a.adv64[row];
        if (a.normalize) x = (x - mean) / (sd + 1e-8);
        if (a.positive) x = (x - mn) + 1e-8;
        a.adv32[row] = (float)x;
    }
}

// ---------------------------------------------------------------------------------------------
// Wide observations (obs_dim > 32, i.e. more than 68 features: Ant's 2*111 + 4 = 226).  k_gram<NBLK> stages raw
// observation rows of at most 32 floats per wave and k_fit keeps two (D+1)^2 matrices in LDS; beyond that:
//
// k_gram_wide: the upper-triangular 16x16 block pairs of [Phi R]^T [Phi R] no longer fit one wave's registers
// (120 pairs at D = 226), so the workgroup shares the feature tile: 8 waves, wave w owns a contiguous eighth of the pair list
// (<= GRAMW_PPW of them) over ALL rows of the work item -- no cross-wave reduction, each partial block is written by
// its owner.  Rounds of GRAMW_ROWS rows; same partial layout as k_gram ([NPAIR][256] doubles per work item).
// grid = work items (table 0) x pair slices, block = 512.  smem: the feature tile, ROWS * FS doubles.
// ---------------------------------------------------------------------------------------------
#define GRAMW_PPW 20     // pairs per wave: 8 * 20 >= 17 * 18 / 2 (NBLK <= 17, D <= 271) in one workgroup; more blocks: the pair list is
                         // cut into gridDim.y slices of at most 160 (gramw_slices), one workgroup per work item and slice
PROMP_HD int gramw_fs(int NBLK) { return (NBLK % 2 == 1) ? 16 * NBLK : 16 * NBLK + 16; }
PROMP_HD size_t gramw_smem(int NBLK, int O, int rows) {
    (void)O;
    return sizeof(double) * (size_t)(rows * gramw_fs(NBLK));      // the feature tile
}
// rows per round: 64 where the feature tile + raw observations fit the 160 KB of LDS (Ant: 151 KB), else 32, else 16
PROMP_HD int gramw_rows(int NBLK, int O) {
    return gramw_smem(NBLK, O, 64) <= 160 * 1024 ? 64 : gramw_smem(NBLK, O, 32) <= 160 * 1024 ? 32 : 16;
}
PROMP_HD int gramw_slices(int NBLK) { return (NBLK * (NBLK + 1) / 2 + 8 * GRAMW_PPW - 1) / (8 * GRAMW_PPW); }

__global__ void __launch_bounds__(512, 2) k_gram_wide(SampleArgs a, int NBLK, int GRAMW_ROWS) {
    PROMP_SMEM_DECL;
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6), i16 = lane & 15, kk = lane >> 4;
    const int FS = gramw_fs(NBLK), NPAIR = NBLK * (NBLK + 1) / 2, NC = 16 * NBLK;
    double* Phi = (double*)PROMP_SMEM_PTR;
    const WorkItem wk = a.work[blockIdx.x];
    const int O = a.O, D = a.D;
    // this wave's pairs: a contiguous range of the row-major list (bi, bj >= bi); offsets of the two 16-column blocks inside a
    // feature row (wave-uniform)
    // (gridDim.y > 1: this workgroup's slice of the list first)
    const int npy = (NPAIR + (int)gridDim.y - 1) / (int)gridDim.y, py0 = npy * (int)blockIdx.y;
    const int pend = py0 + npy < NPAIR ? py0 + npy : NPAIR;
    const int ppw = (npy + 7) / 8, p0 = py0 + w * ppw;
    int ca[GRAMW_PPW], cb[GRAMW_PPW];
#pragma unroll
    for (int j = 0; j < GRAMW_PPW; ++j) {
        const int p = p0 + j;
        int bi = 0, rem = (j < ppw && p < pend) ? p : 0;
        while (rem >= NBLK - bi) {
            rem -= NBLK - bi;
            ++bi;
        }
        ca[j] = 16 * bi;
        cb[j] = 16 * (bi + rem);
    }
    f64x4 acc[GRAMW_PPW];
#pragma unroll
    for (int j = 0; j < GRAMW_PPW; ++j) acc[j] = zero4d();
    // The feature tile is written straight from the loaded observations (element e of the [ROWS][O] chunk is observation e mod O
    // of row e / O: columns c and O + c), 16 loads per thread in flight; threads 0 .. ROWS - 1 add their row's time features, the
    // constant and the target.  Columns past D + 1 are zeroed once.  (Until round 4 the rows were staged raw in LDS and every
    // feature column went through a branch tree: k_gram's 32.8 k of 52 k cycles.)
    for (int e = tid; e < GRAMW_ROWS * FS; e += 512) Phi[e] = 0.0;
    const float rO = 1.0f / (float)(O > 0 ? O : 1);
    const int qt = (a.kind == BASE_LINFEAT) ? 2 * O : 0;        // first of the four time columns (tau, tau^2, tau^3, 1)
    for (int base = wk.row_begin; base < wk.row_end; base += GRAMW_ROWS) {
        const int nrows = (wk.row_end - base) < GRAMW_ROWS ? (wk.row_end - base) : GRAMW_ROWS;
        __syncthreads();   // the previous round's MFMAs are done with Phi (first round: the zeros are in place)
        if (a.kind == BASE_LINFEAT) {
            const int lim = nrows * O, tot = GRAMW_ROWS * O;
            for (int e0 = tid; e0 < tot; e0 += 512 * 16) {
                float x[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int e = e0 + 512 * u;
                    x[u] = a.obs[(long long)base * O + (e < lim ? e : 0)];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int e = e0 + 512 * u;
                    if (e < tot) {
                        const int r = (int)(((float)e + 0.5f) * rO), c = e - r * O;
                        // (rows past the chunk's last become zeros.)  the reference squares in the observations' own dtype, then promotes
                        const float oc = e < lim ? fminf(fmaxf(x[u], -10.f), 10.f) : 0.f;
                        Phi[r * FS + c] = (double)oc;
                        Phi[r * FS + O + c] = (double)(oc * oc);
                    }
                }
            }
        }
        if (tid < GRAMW_ROWS) {
            const bool rv = tid < nrows;
            const int r = rv ? tid : 0;
            const double t = a.ret64[base + r];
            const double tau = rv ? (double)a.row_t[base + r] / 100.0 : 0.0;
            double* pr = Phi + tid * FS;
            pr[qt] = tau;
            pr[qt + 1] = tau * tau;
            pr[qt + 2] = tau * tau * tau;
            pr[qt + 3] = rv ? 1.0 : 0.0;
            pr[D] = rv ? t : 0.0;
        }
        __syncthreads();
        // The products, software-pipelined: the operands of the NEXT group of four pairs are requested before the current group's
        // four matrix instructions issue (4 x 64 cycles of FP64 MFMA cover the LDS round trip).  Loading a group, waiting, then
        // issuing its products left the matrix pipe idle two thirds of the time (PMC: SQ_VALU_MFMA_BUSY_CYCLES 33 % of the
        // kernel, SQ_WAIT_ANY 50 % of the wave cycles).  Pairs beyond the wave's share read a valid address and are not issued.
        {
            constexpr int NG = GRAMW_PPW / 4;
            const int nsteps = GRAMW_ROWS / 4;
            double xa[4], xb[4];
            const double* row = Phi + kk * FS + i16;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xa[u] = row[ca[u]];
                xb[u] = row[cb[u]];
            }
#pragma unroll 1
            for (int st = 0; st < nsteps; ++st) {
                const double* rnext = Phi + (4 * (st + 1 < nsteps ? st + 1 : st) + kk) * FS + i16;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    double na[4], nb[4];
                    const double* src = (g + 1 < NG) ? row : rnext;
                    const int gn = (g + 1 < NG) ? g + 1 : 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        na[u] = src[ca[4 * gn + u]];
                        nb[u] = src[cb[4 * gn + u]];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = 4 * g + u;
                        if (j < ppw && p0 + j < pend) acc[j] = mfma16d(xa[u], xb[u], acc[j]);      // (wave-uniform)
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        xa[u] = na[u];
                        xb[u] = nb[u];
                    }
                    sched_fence();
                }
                row = rnext;
            }
        }
    }
    double* out = a.gram_partials + (long long)blockIdx.x * (NPAIR * 256);
#pragma unroll
    for (int j = 0; j < GRAMW_PPW; ++j) {
        const int p = p0 + j;
        if (j < ppw && p < pend)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[p * 256 + (kk + 4 * r) * 16 + i16] = acc[j][r];
    }
}

// ---------------------------------------------------------------------------------------------
// k_gram_tiled (round 5, second half): the same product as k_gram_wide with the operands of the FP64 matrix instruction REUSED in
// registers.  k_gram_wide reads two LDS operands per v_mfma_f64_16x16x4_f64 (its waves own arbitrary runs of the pair list, so the
// shared row block of a run is fetched again for every pair): 40 ds_read_b64 per 15 products and wave at Ant's width, two thirds
// of the LDS bandwidth of the CU when the matrix pipe is full -- the pipe measured 44 % busy.  Here a wave owns a SQUARE of
// TB x TB column blocks (bands of TB = 3 blocks, 48 feature columns; band pair bi <= bj): 3 + 3 operand reads feed 9 products, at
// immediate offsets from two lane addresses.  Squares on the diagonal issue their upper triangle (6 products from 3 reads).  The
// two shapes are two instances of the step loop: products behind wave-uniform branches, or more loop shapes (short last bands),
// made the register allocator spill accumulators -- the block count is padded to whole bands instead (the padding columns are
// zero in the tile and are never written out).  out[] keeps k_gram_wide's layout ([pair][256] per work item, pair = row-major
// index of the upper block triangle), so k_gram_sum_wide / the fits do not change.
//   * Ant's 15 blocks are 5 bands = 15 squares = 15 waves of ONE workgroup (10 x 9 + 5 x 6 = 120 products per k-step, the
//     unpadded count); the host shares them out over the SIMDs by cost (GramtMap; wave w runs on SIMD w mod 4).  More squares:
//     slices of NWV along gridDim.y (Humanoid's 48 blocks: 136 squares, 9 slices).
//   * The feature tile is double-buffered where two tiles fit LDS (Ant: 2 x 32 rows x 240 columns = 123 KB): the observations
//     of round r + 1 are requested before the products of round r and written into the other tile behind them -- one barrier
//     per round, the global latency and the tile build off the matrix pipe's path.  Wider tiles (Humanoid: 16 rows x 784) are
//     single: the request is still issued a round ahead, the write sits between two barriers.
//   * A partial last round issues only the k-steps that hold rows.
// grid = work items (table 0) x slices, block = 64 NWV.  ROWS / DB: gramt_cfg; smem: gramt_smem(NBLK, ROWS, DB).
// ---------------------------------------------------------------------------------------------
#define GRAMT_MIN_NBLK 13
#define GRAMT_TB 3
#define GRAMT_NWV 16
#define GRAMT_NLD 8
struct GramtMap {        // one-slice launches: wave -> square (255: none) and which part of a diagonal square (GRAMT_*)
    unsigned char rect[16], part[16];
};
enum { GRAMT_FULL = 0, GRAMT_DIAG = 1, GRAMT_DIAG_TOP = 2, GRAMT_DIAG_REST = 3 };
PROMP_HD int gramt_nb(int NBLK) { return (NBLK + GRAMT_TB - 1) / GRAMT_TB; }
PROMP_HD int gramt_fs(int NBLK) {       // 16 x odd: the four k-rows of a step land on disjoint banks
    const int nc = GRAMT_TB * gramt_nb(NBLK);
    return (nc % 2 == 1) ? 16 * nc : 16 * nc + 16;
}
PROMP_HD int gramt_nrect(int NBLK) { return gramt_nb(NBLK) * (gramt_nb(NBLK) + 1) / 2; }
// rows per round and single / double tile: two tiles of 32 or 16 rows where they fit LDS (and a round's observations the
// threads' request registers: cap = NT * NLD elements) -- the build of round r + 1 then runs beside the products of round r
// behind ONE barrier per round; `single` (PROMP_GRAMT_SINGLE=1, the A/B switch) or nothing fitting twice: one tile of 32 / 16 rows,
// the build between two barriers.
PROMP_HD void gramt_cfg(int NBLK, int O, int cap, bool single, int* rows, int* db) {
    const size_t row_bytes = sizeof(double) * (size_t)gramt_fs(NBLK), lds = 160 * 1024;
    if (!single)
        for (int r = 32; r >= 16; r >>= 1)      // (two tiles of 8 rows measured slower than one of 16 at Humanoid's width: 2.37 vs 2.21 ms)
            if (2 * r * row_bytes <= lds && r * O <= cap) { *rows = r; *db = 1; return; }
    *rows = (32 * row_bytes <= lds && 32 * O <= cap) ? 32 : 16;
    *db = 0;
}
PROMP_HD size_t gramt_smem(int NBLK, int rows, int db) { return sizeof(double) * (size_t)((db ? 2 : 1) * rows * gramt_fs(NBLK)); }

// which of a square's TB x TB products a wave of shape SHAPE issues: all (off the diagonal); the upper triangle jj >= ii (a
// diagonal square); the triangle's first row / its other rows (a diagonal square shared by two waves: with a wave to spare the
// host splits one so that the four SIMDs carry the same number of products)
template <int SHAPE>
PROMP_HD PROMP_CX bool gramt_has(int ii, int jj) {
    return SHAPE == GRAMT_FULL ? true : SHAPE == GRAMT_DIAG ? jj >= ii : SHAPE == GRAMT_DIAG_TOP ? ii == 0 : (ii >= 1 && jj >= ii);
}
// the k-steps of one round for one square: pa / pb = this lane's element of the first block of the row / column band in k-row
// kk of the tile; FS4 = 4 rows of the tile.  On the diagonal the column band IS the row band: one set of operands.
template <int TB, int SHAPE>
PROMP_DEV void gramt_steps(f64x4 (&acc)[TB][TB], const double* pa, const double* pb, int nst, int FS4) {
    constexpr bool DIAG = SHAPE != GRAMT_FULL;
#pragma unroll 1
    for (int st = 0; st < nst; ++st) {
        double fa[TB], fb[TB];
#pragma unroll
        for (int u = 0; u < TB; ++u) {
            fa[u] = pa[16 * u];
            fb[u] = DIAG ? fa[u] : pb[16 * u];
        }
#pragma unroll
        for (int ii = 0; ii < TB; ++ii)
#pragma unroll
            for (int jj = 0; jj < TB; ++jj)
                if (gramt_has<SHAPE>(ii, jj)) acc[ii][jj] = mfma16d(fa[ii], fb[jj], acc[ii][jj]);
        pa += FS4;
        pb += FS4;
    }
}

// One wave's whole walk over the work item's rows for its square (bi, bj); SHAPE: GRAMT_*.  The shapes are separate instances of
// the WHOLE walk, accumulators included: with one set of accumulators around two step loops the compiler gave each loop its own
// registers for them (72 + 48 of 128) and spilled.  Every wave passes the same barriers whatever its shape.
template <int TB, int NWV, int NLD, int SHAPE>
PROMP_DEV void gramt_walk(const SampleArgs& a, int NBLK, int ROWS, int DB, double* Phi, bool active, int bi, int bj) {
    constexpr int NT = 64 * NWV;
    const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, kk = lane >> 4;
    const int FS = gramt_fs(NBLK);
    const int O = a.O, D = a.D;
    const WorkItem wk = a.work[blockIdx.x];
    f64x4 acc[TB][TB];
#pragma unroll
    for (int ii = 0; ii < TB; ++ii)
#pragma unroll
        for (int jj = 0; jj < TB; ++jj) acc[ii][jj] = zero4d();

    // The tile build, as k_gram_wide's: element e of the round's [ROWS][O] chunk of observations is observation e mod O of row
    // e / O and goes to columns c and O + c; threads 0 .. ROWS - 1 add their row's time features, the constant and the target.
    // Split in two: the requests (into x[], tgt, rt) and, a round of products later, the writes.
    const float rO = 1.0f / (float)(O > 0 ? O : 1);
    const int qt = (a.kind == BASE_LINFEAT) ? 2 * O : 0;        // first of the four time columns (tau, tau^2, tau^3, 1)
    const int tot = ROWS * O;
    float x[NLD];
    double tgt = 0.0;
    int rt = 0;
    auto request = [&](int base, int nrows) {
        if (a.kind == BASE_LINFEAT) {
            const int lim = nrows * O;
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int e = tid + NT * u;
                x[u] = a.obs[(long long)base * O + (e < lim ? e : 0)];
            }
        }
        if (tid < ROWS) {
            const int r = tid < nrows ? tid : 0;
            tgt = a.ret64[base + r];
            rt = a.row_t[base + r];
        }
    };
    auto build = [&](double* buf, int nrows) {
        if (a.kind == BASE_LINFEAT) {
            const int lim = nrows * O;
            // (the positions are recomputed every round: kept across the products they would cost 2 NLD registers)
            const int oz = opaque_zero();
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int e = tid + NT * u + oz;
                if (e < tot) {
                    const int r = (int)(((float)e + 0.5f) * rO), c = e - r * O;
                    // (rows past the chunk's last become zeros.)  the reference squares in the observations' own dtype, then promotes
                    const float oc = e < lim ? fminf(fmaxf(x[u], -10.f), 10.f) : 0.f;
                    buf[r * FS + c] = (double)oc;
                    buf[r * FS + O + c] = (double)(oc * oc);
                }
                sched_fence();      // one element at a time: interleaved, the NLD elements' temporaries push accumulators into scratch
            }
        }
        if (tid < ROWS) {
            const bool rv = tid < nrows;
            const double tau = rv ? (double)rt / 100.0 : 0.0;
            double* pr = buf + tid * FS;
            pr[qt] = tau;
            pr[qt + 1] = tau * tau;
            pr[qt + 2] = tau * tau * tau;
            pr[qt + 3] = rv ? 1.0 : 0.0;
            pr[D] = rv ? tgt : 0.0;
        }
    };

    for (int e = tid; e < (DB ? 2 : 1) * ROWS * FS; e += NT) Phi[e] = 0.0;      // columns past D + 1 stay zero
    __syncthreads();
    {
        const int n0 = (wk.row_end - wk.row_begin) < ROWS ? (wk.row_end - wk.row_begin) : ROWS;
        request(wk.row_begin, n0);
        build(Phi, n0);
    }
    __syncthreads();
    int cur = 0;
    const int lofs = kk * FS + i16;
    for (int base = wk.row_begin; base < wk.row_end; base += ROWS) {
        const int nrows = (wk.row_end - base) < ROWS ? (wk.row_end - base) : ROWS;
        const int nbase = base + ROWS;
        const bool more = nbase < wk.row_end;
        const int nnext = (wk.row_end - nbase) < ROWS ? (wk.row_end - nbase) : ROWS;
        if (more) request(nbase, nnext);
        const double* buf = Phi + cur * ROWS * FS;
        if (active) {
            const int nst = (nrows + 3) >> 2;
            const double* pa = buf + lofs + 16 * TB * bi;
            const double* pb = buf + lofs + 16 * TB * bj;
            gramt_steps<TB, SHAPE>(acc, pa, pb, nst, 4 * FS);
        }
        if (DB) {
            if (more) build(Phi + (cur ^ 1) * ROWS * FS, nnext);    // nobody reads that tile before the barrier below
            __syncthreads();
            cur ^= 1;
        } else {
            __syncthreads();                                        // the products of this round are done with the tile
            if (more) build(Phi, nnext);
            __syncthreads();
        }
    }
    const int NPAIR = NBLK * (NBLK + 1) / 2;
    double* out = a.gram_partials + (long long)blockIdx.x * (NPAIR * 256);
#pragma unroll
    for (int ii = 0; ii < TB; ++ii)
#pragma unroll
        for (int jj = 0; jj < TB; ++jj) {
            const int gi = TB * bi + ii, gj = TB * bj + jj;
            if (active && gramt_has<SHAPE>(ii, jj) && gj < NBLK) {
                const int p = gi * NBLK - gi * (gi - 1) / 2 + (gj - gi);
#pragma unroll
                for (int r = 0; r < 4; ++r) out[p * 256 + (kk + 4 * r) * 16 + i16] = acc[ii][jj][r];
            }
        }
}

template <int TB, int NWV, int NLD>
__global__ void __launch_bounds__(64 * NWV) k_gram_tiled(SampleArgs a, int NBLK, GramtMap map, int ROWS, int DB) {
    PROMP_SMEM_DECL;
    const int w = wave_uniform((int)threadIdx.x >> 6);
    const int NB = gramt_nb(NBLK), NR = NB * (NB + 1) / 2;
    // this wave's square: band pair bi <= bj of the row-major list
    const int rect = wave_uniform(gridDim.y == 1 ? (int)map.rect[w] : (int)blockIdx.y * NWV + w);
    const bool active = rect < NR;
    int bi = 0, rem = active ? rect : 0;
    while (rem >= NB - bi) {
        rem -= NB - bi;
        ++bi;
    }
    const int bj = bi + rem;
    const int part = wave_uniform(gridDim.y == 1 ? (int)map.part[w] : GRAMT_DIAG);
    double* Phi = (double*)PROMP_SMEM_PTR;
    if (bi != bj) gramt_walk<TB, NWV, NLD, GRAMT_FULL>(a, NBLK, ROWS, DB, Phi, active, bi, bj);
    else if (part == GRAMT_DIAG_TOP) gramt_walk<TB, NWV, NLD, GRAMT_DIAG_TOP>(a, NBLK, ROWS, DB, Phi, active, bi, bj);
    else if (part == GRAMT_DIAG_REST) gramt_walk<TB, NWV, NLD, GRAMT_DIAG_REST>(a, NBLK, ROWS, DB, Phi, active, bi, bj);
    else gramt_walk<TB, NWV, NLD, GRAMT_DIAG>(a, NBLK, ROWS, DB, Phi, active, bi, bj);
}

// ---------------------------------------------------------------------------------------------
// k_fit_wide: the same solve as k_fit when the (D+1)^2 matrices do not fit in LDS (Ant: D = 226).  G and the work matrix live in
// global memory (L2-resident, 0.4 MB per task).  Right-looking blocked Cholesky, 32-column panels through LDS, the right-hand
// side carried along as row D (k_fit's forward solve for free):
//   1. the panel's 32 x 32 diagonal block is factorised by ONE wave in registers (a row per lane, pivots and multipliers by
//      v_readlane: k_fit_wave's column step -- 32 dependent steps with no barrier and no memory access on the chain);
//   2. the rows below it (and the right-hand-side row) are solved against that block one row per thread (independent rows);
//   3. the rank-32 update of the trailing matrix runs on the FP64 matrix cores: v_mfma_f64_16x16x4_f64 over the 16 x 16 tiles of
//      the lower triangle, all waves, operands straight from the panel in LDS.
// The back substitution walks the factor in 32-row blocks: one wave solves the block's triangle in registers, all threads apply
// its 32 solved entries to the rows above.  Round 3's version took two workgroup barriers per COLUMN (226 x 2, plus 226 in the
// back substitution) and updated the trailing matrix with scalar FMAs: 676 us per launch at Ant's size; this one takes four
// barriers per PANEL.  Same elimination order and arithmetic as k_fit up to the reciprocal square root k_fit_wave also uses
// (results agree to rounding, not bit for bit).
// grid = tasks, block = FITW_NT.  smem: fitw_smem(D).
// ---------------------------------------------------------------------------------------------
#define FITW_NB 32         // panel width where the panel fits LDS (D <= ~580); wider matrices take 16-column panels
#define FITW_NT 512        // 8 waves: two per SIMD, 256 registers each (a row of the diagonal block / of the solve lives in 64 of them)
PROMP_HD size_t fitw_smem(int D, int nb) {
    const size_t DA = D + 1, panel = (DA + 16) * (size_t)(nb + 1);       // (16 spare rows: the last 16-row tile of the update reads zeros)
    return sizeof(double) * (panel + 3 * DA + nb + 2);
}
PROMP_HD int fitw_nb(int D) { return fitw_smem(D, FITW_NB) <= 160 * 1024 ? FITW_NB : 16; }

// The task's partial Gram blocks summed in workgroup order and scattered into the symmetric matrix G -- and G + reg I into the
// work matrix of the first factorisation attempt -- by the WHOLE chip (grid = tasks x fitw_sum_split(NBLK)): inside k_fit_wide the sum
// ran on one compute unit per task (40 of 256 busy, a quarter of that kernel's time at Ant's size).
// scratch: [tasks][2][(D+1)^2].  block = 256.
// (the split grows with the matrix: Humanoid's 1176 blocks of 256 entries per task are 617 MB of partial blocks per launch, and with
//  8 workgroups per task the loads in flight -- not the memory system -- set the pace; every entry is still one thread's sum in workgroup order)
PROMP_HD int fitw_sum_split(int NBLK) { return NBLK >= 32 ? 32 : NBLK >= 13 ? 16 : 8; }
__global__ void __launch_bounds__(256) k_gram_sum_wide(SampleArgs a, int NBLK, double* scratch, int FITW_SUM_SPLIT) {
    const int D = a.D, DA = D + 1;
    const int task = blockIdx.x / FITW_SUM_SPLIT, part = blockIdx.x % FITW_SUM_SPLIT;
    double* G = scratch + (size_t)task * 2 * DA * DA;
    double* Wm = G + (size_t)DA * DA;
    const int NPAIR = NBLK * (NBLK + 1) / 2;
    const int wg0 = a.task_wg_offsets[task], wg1 = a.task_wg_offsets[task + 1];
    for (int e = part * 256 + threadIdx.x; e < NPAIR * 256; e += 256 * FITW_SUM_SPLIT) {
        // eight partial blocks per trip, all requested before the first is added (as k_fit_wave)
        double s = 0.0;
        for (int base = wg0; base < wg1; base += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int wg = base + u < wg1 ? base + u : wg1 - 1;
                v[u] = a.gram_partials[(long long)wg * (NPAIR * 256) + e];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (base + u < wg1) ? v[u] : 0.0;
        }
        int p = e >> 8, bi = 0, rem = p;
        while (rem >= NBLK - bi) {
            rem -= NBLK - bi;
            ++bi;
        }
        const int bj = bi + rem;
        const int row = 16 * bi + ((e & 255) >> 4), col = 16 * bj + (e & 15);
        if (row < DA && col < DA) {
            const double diag = (row == col && row < D) ? a.reg : 0.0;
            G[(size_t)row * DA + col] = s;
            Wm[(size_t)row * DA + col] = s + diag;
            if (bi != bj) {
                G[(size_t)col * DA + row] = s;
                Wm[(size_t)col * DA + row] = s;
            }
        }
    }
}

// Back substitution L^T w = z over the factor's NB-row blocks from the bottom up (k_fit_wide, k_fitw_back): wave 0 solves the block's
// triangle in registers (lane t holds y[kb + t] and column t of the transposed triangle), the other waves then apply the block's
// solved entries to the rows above, one row per thread.  yv = z on entry (LDS, written by the caller without a barrier), rdv =
// 1 / diag(L) (LDS, likewise), the result goes to wv (LDS); tri: 2 x NB x (NB + 1) doubles of LDS.
// The factor lives in global memory (L2) and is final, so nothing taken from there depends on the solve: a thread requests the NB
// factor entries of its row BEFORE wave 0 solves the block, and the NEXT block's triangle is requested at the same point (by waves
// 1.., a few entries per thread, parked in LDS behind the first barrier: in wave 0's registers a second triangle spilled) -- a block
// costs one solve (NB dependent steps) instead of two round trips to L2 plus the solve (21 % of k_fit_wide's time at Ant's 8 blocks;
// 48 blocks at Humanoid's width).  Same operations on the same values in the same order as the loop it replaces.
template <int NB>
PROMP_DEV void fitw_back_substitute(const double* Wm, int D, int DA, double* yv, double* wv, const double* rdv, double* tri, int tid,
                                    int lane, int w) {
    constexpr int NT = FITW_NT, NH = NT - 64, TS = NB + 1, NTV = (NB * NB + NH - 1) / NH;
    const int row = tid - 64;                 // (wave 0 solves; the rows above and the prefetch belong to the other waves)
    int kb = ((D - 1) / NB) * NB, cur = 0;
    {
        const int nb = (D - kb) < NB ? (D - kb) : NB;
        for (int e = row; e >= 0 && e < NB * NB; e += NH) {
            const int j = e / NB, t = e - j * NB;
            if (j < nb && j > t) tri[j * TS + t] = Wm[(size_t)(kb + j) * DA + kb + t];
        }
    }
    __syncthreads();
    for (; kb >= 0; kb -= NB) {
        const int nb = (D - kb) < NB ? (D - kb) : NB;
        const bool mine = row >= 0 && row < kb;
        double lv[NB], tv[NTV];
        if (mine) {
#pragma unroll
            for (int r = 0; r < NB; ++r) lv[r] = Wm[(size_t)(kb + (r < nb ? r : 0)) * DA + row];     // all in flight together
        }
        if (row >= 0 && kb >= NB) {           // the next block's triangle (every block above the last one is full)
            const int kn = kb - NB;
#pragma unroll
            for (int u = 0; u < NTV; ++u) {
                const int e = row + u * NH, j = e / NB, t = e - j * NB;
                tv[u] = (e < NB * NB && j > t) ? Wm[(size_t)(kn + j) * DA + kn + t] : 0.0;
            }
        }
        if (w == 0) {
            const int t = lane < nb ? lane : nb - 1;
            const double* T = tri + cur * NB * TS;
            double W[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) W[j] = (j < nb && j > t) ? T[j * TS + t] : 0.0;
            double y = yv[kb + t];
            const double rd = rdv[kb + t];
            double wsol = 0.0;
#pragma unroll
            for (int j = NB - 1; j >= 0; --j) {
                if (j < nb) {
                    const double wj = readlane_f64(y, j) * readlane_f64(rd, j);
                    if (lane == j) wsol = wj;
                    if (lane < j) y -= W[j] * wj;
                }
            }
            if (lane < nb) wv[kb + lane] = wsol;
        }
        __syncthreads();
        // the rows above: y[i] -= sum_r L[kb + r][i] w[kb + r]
        if (mine) {
            double s = yv[row];
#pragma unroll
            for (int r = 0; r < NB; ++r) s -= (r < nb) ? lv[r] * wv[kb + r] : 0.0;
            yv[row] = s;
        }
        for (int i = row + NH; row >= 0 && i < kb; i += NH) {         // (more than NT - 64 rows above: Humanoid's first blocks)
            double l2[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) l2[r] = Wm[(size_t)(kb + (r < nb ? r : 0)) * DA + i];
            double s = yv[i];
#pragma unroll
            for (int r = 0; r < NB; ++r) s -= (r < nb) ? l2[r] * wv[kb + r] : 0.0;
            yv[i] = s;
        }
        if (row >= 0 && kb >= NB) {
            double* Tn = tri + (cur ^ 1) * NB * TS;
#pragma unroll
            for (int u = 0; u < NTV; ++u) {
                const int e = row + u * NH, j = e / NB, t = e - j * NB;
                if (e < NB * NB && j > t) Tn[j * TS + t] = tv[u];
            }
        }
        cur ^= 1;
        __syncthreads();
    }
}

// only_bad != nullptr: the launch behind the one-launch-per-phase factorisation below -- tasks whose flag is 0 are done; the others
// (a NaN in the solution: the ridge term was too small) start over at the second attempt, reg x 10, exactly as they would have here.
template <int NB>
__global__ void __launch_bounds__(FITW_NT) k_fit_wide(SampleArgs a, int NBLK, double* scratch, const int* only_bad) {
    constexpr int PS = NB + 1, LOG_NB = NB == 32 ? 5 : 4;
    PROMP_SMEM_DECL;
#ifdef PROMP_DEV_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tph = promp_clock();
#define FITW_PHASE(i) do { const unsigned long long now_ = promp_clock(); ph[i] += now_ - tph; tph = now_; } while (0)
#else
#define FITW_PHASE(i) do { } while (0)
#endif
    const int D = a.D, DA = D + 1, NT = FITW_NT;
    const int tid = threadIdx.x, task = blockIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
    const int i16 = lane & 15, kk = lane >> 4;
    double* Pn = (double*)PROMP_SMEM_PTR;     // panel [DA - k0 (+ 16)][33]
    double* yv = Pn + (size_t)(DA + 16) * PS;
    double* wv = yv + DA;
    double* dg = wv + DA;
    double* rdp = dg + DA;                    // 1 / L[c][c] of the current panel's diagonal block
    int* flag = (int*)(rdp + NB);
    double* G = scratch + (size_t)task * 2 * DA * DA;
    double* Wm = G + (size_t)DA * DA;
    // (G, and G + reg I as the work matrix of the first attempt, were written by k_gram_sum_wide)
    const float rDA = 1.0f / (float)DA;
    if (only_bad != nullptr && !only_bad[task]) return;          // (uniform over the workgroup)
    const int first_attempt = only_bad != nullptr ? 1 : 0;
    double reg = first_attempt ? a.reg * 10.0 : a.reg;
    for (int attempt = first_attempt; attempt < 5; ++attempt) {
        if (attempt > 0) {            // a retry starts over from G with the larger ridge term
            for (int e = tid; e < DA * DA; e += NT) {
                const int i = (int)(((float)e + 0.5f) * rDA), j = e - i * DA;
                Wm[e] = G[e] + ((i == j && i < D) ? reg : 0.0);
            }
        }
        __syncthreads();
        FITW_PHASE(1);
        for (int k0 = 0; k0 < D; k0 += NB) {
            const int nb = (D - k0) < NB ? (D - k0) : NB;   // columns of this panel
            const int nr = DA - k0;                                  // rows k0..D (the last one is the right-hand side)
            for (int e = tid; e < (nr + 16) * NB; e += NT) {
                const int i = e >> LOG_NB, c = e & (NB - 1);
                Pn[i * PS + c] = (c < nb && i < nr) ? Wm[(size_t)(k0 + i) * DA + k0 + c] : 0.0;
            }
            __syncthreads();
            FITW_PHASE(2);
            if (w == 0) {
                // ---- the diagonal block, one wave, rows in registers (lanes >= nb shadow the last row: finite, never stored)
                const int row = lane < nb ? lane : nb - 1;
                double W[NB];
                double rdg = 1.0, dgv = 0.0;
#pragma unroll
                for (int k = 0; k < NB; ++k) W[k] = Pn[row * PS + k];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if (j < nb) {     // (wave-uniform)
                        const double dj = readlane_f64(W[j], j), rs = rsqrt(dj), piv = dj * rs;
                        W[j] = (lane == j) ? piv : W[j] * rs;
                        rdg = (lane == j) ? rs : rdg;
                        dgv = (lane == j) ? piv : dgv;       // (W[lane] read at the end would index the array by a runtime value: the
                                                             //  whole row then lives in scratch memory, loop included)
#pragma unroll
                        for (int k = j + 1; k < NB; ++k) W[k] -= W[j] * readlane_f64(W[j], k);     // lane k holds L[k][j] in W[j]
                    }
                }
                if (lane < nb) {
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        if (k <= lane) Pn[lane * PS + k] = W[k];
                    dg[k0 + lane] = dgv;
                    rdp[lane] = rdg;
                }
            }
            __syncthreads();
            FITW_PHASE(3);
            // ---- the rows below: x L^T = a, one row per thread (forward substitution over the block's columns)
            for (int rr = tid; rr < nr - nb; rr += NT) {       // (one trip up to 512 rows below the block)
                double* xr = Pn + (size_t)(nb + rr) * PS;
                double x[NB];
#pragma unroll
                for (int c = 0; c < NB; ++c) x[c] = xr[c];
                // column by column, right-looking: once x[c] is final the later entries take their x[c] L[c2][c] at once -- 31 - c
                // independent multiply-adds (the left-looking form is one dependent chain of c per entry).  L[c2][c] is the same
                // address in every lane: a broadcast.  In a last panel narrower than 32 the rows / columns >= nb of the block hold
                // other rows' data: finite junk that only reaches entries >= nb, which are stored as 0.
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    if (c < nb) {
                        x[c] *= rdp[c];
#pragma unroll
                        for (int c2 = c + 1; c2 < NB; ++c2) x[c2] -= x[c] * Pn[c2 * PS + c];
                    }
                }
#pragma unroll
                for (int c = 0; c < NB; ++c) xr[c] = (c < nb) ? x[c] : 0.0;
            }
            __syncthreads();
            FITW_PHASE(4);
            // write the factored panel back (strictly-lower part and the right-hand-side row)
            for (int e = tid; e < nr * NB; e += NT) {
                const int i = e >> LOG_NB, c = e & (NB - 1);
                if (c < nb && i > c) Wm[(size_t)(k0 + i) * DA + k0 + c] = Pn[i * PS + c];
            }
            // ---- rank-nb update of the trailing matrix on the FP64 matrix cores: rows k1..D, columns k1..D-1, the tiles of the
            //      lower triangle, wave w takes tiles w, w + 16, ...
            const int k1 = k0 + nb, tr = DA - k1, tc = D - k1;
            if (tc > 0) {
                const int nti = (tr + 15) >> 4, ntj = (tc + 15) >> 4;
                const int ntile = nti * ntj;                    // (tiles above the diagonal are skipped below)
                for (int tix = w; tix < ntile; tix += FITW_NT / 64) {
                    const int bi = tix / ntj, bj = tix - bi * ntj;
                    if (bj > bi) continue;                     // wave-uniform
                    const double* pa = Pn + (size_t)(nb + 16 * bi + i16) * PS + kk;
                    const double* pb = Pn + (size_t)(nb + 16 * bj + i16) * PS + kk;
                    // the entries this lane will update, requested before the products (clamped addresses, masked at the store)
                    double* dst[4];
                    double old[4];
                    bool ok[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int io = 16 * bi + kk + 4 * r, jo = 16 * bj + i16;
                        ok[r] = io < tr && jo < tc && jo <= io;
                        dst[r] = Wm + (size_t)(k1 + (ok[r] ? io : 0)) * DA + k1 + (ok[r] ? jo : 0);
                        old[r] = *dst[r];
                    }
                    f64x4 acc = zero4d();
#pragma unroll
                    for (int sidx = 0; sidx < NB / 4; ++sidx) acc = mfma16d(pa[4 * sidx], pb[4 * sidx], acc);
                    // D: col = lane & 15, row = (lane >> 4) + 4 r
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ok[r]) *dst[r] = old[r] - acc[r];
                }
            }
            __syncthreads();
            FITW_PHASE(5);
        }
        // ---- back substitution L^T w = z, z = row D of the factor, 32 rows of L at a time (the reciprocals of the factor's diagonal
        //      take the diagonal's place: one division per thread here instead of one per block on wave 0's chain)
        for (int e = tid; e < D; e += NT) {
            yv[e] = Wm[(size_t)D * DA + e];
            dg[e] = 1.0 / dg[e];
        }
        fitw_back_substitute<NB>(Wm, D, DA, yv, wv, dg, Pn, tid, lane, w);     // (the panel's LDS is free: two triangles)
        FITW_PHASE(6);
        if (tid == 0) *flag = 0;
        __syncthreads();
        for (int e = tid; e < D; e += NT)
            if (wv[e] != wv[e]) *flag = 1;
        __syncthreads();
        const int bad = *flag;
        __syncthreads();
        if (!bad) break;
        reg *= 10.0;
    }
    for (int e = tid; e < D; e += NT) a.coeffs[(long long)task * a.coeff_stride + e] = wv[e];
#ifdef PROMP_DEV_STAMPS
    if (task == 0 && tid == 0)
        printf("k_fit_wide cycles: entry %llu | retry init %llu | panel load %llu | diagonal block %llu | rows below %llu | store + update %llu | back substitution %llu\n",
               ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6]);
#endif
}

// ---------------------------------------------------------------------------------------------
// The same factorisation with ONE LAUNCH PER PHASE, for matrices so large that one compute unit per task is the wrong shape
// (Humanoid: D = 756, 48 panels of 16 columns; the rank-16 update of a 740 x 740 trailing matrix is 1100 tiles of 16 x 16 per
// panel -- in k_fit_wide 8 waves of ONE CU walk them while 216 CUs idle: 3.5 ms per fit at 40 tasks):
//   k_fitw_panel  (grid = tasks)                    panel k0: diagonal block, rows below, factored panel back into the work matrix
//   k_fitw_update (grid = tasks x FITW_UPD_SPLIT)   trailing matrix -= panel panel^T on the FP64 matrix cores, all CUs
//   ... per panel, then
//   k_fitw_back   (grid = tasks)                    back substitution, NaN check -> coefficients, or bad[task] = 1
//   k_fit_wide(only_bad)                            the (rare) retries with a larger ridge term, unchanged
// Same elimination order and the same arithmetic per entry as k_fit_wide (every trailing entry takes one 16-term MFMA sum per
// panel whichever wave computes it): bit-identical coefficients.  The diagonal of the factor is kept on the work matrix' diagonal.
// ---------------------------------------------------------------------------------------------
#define FITW_UPD_SPLIT 6          // 40 tasks x 6 = 240 workgroups
#define FITW_ML_MIN_D 400         // below: k_fit_wide (8 panels at Ant's 226 columns are a chain of dependent steps, not tile work)
PROMP_HD size_t fitw_panel_smem(int D, int nb) { return sizeof(double) * ((size_t)(D + 1 + 16) * (nb + 1) + nb + 2); }
PROMP_HD size_t fitw_back_smem(int D, int nb) { return sizeof(double) * ((size_t)3 * (D + 1) + 2 * nb * (nb + 1) + 2); }

template <int NB>
__global__ void __launch_bounds__(FITW_NT) k_fitw_panel(SampleArgs a, double* scratch, int k0) {
    constexpr int PS = NB + 1, LOG_NB = NB == 32 ? 5 : 4;
    PROMP_SMEM_DECL;
    const int D = a.D, DA = D + 1, NT = FITW_NT;
    const int tid = threadIdx.x, task = blockIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
    double* Pn = (double*)PROMP_SMEM_PTR;             // [DA - k0][PS]
    double* rdp = Pn + (size_t)(DA + 16) * PS;        // 1 / L[c][c] of the diagonal block
    double* Wm = scratch + (size_t)task * 2 * DA * DA + (size_t)DA * DA;
    const int nb = (D - k0) < NB ? (D - k0) : NB;     // columns of this panel
    const int nr = DA - k0;                           // rows k0..D (the last one is the right-hand side)
    for (int e = tid; e < nr * NB; e += NT) {
        const int i = e >> LOG_NB, c = e & (NB - 1);
        Pn[i * PS + c] = c < nb ? Wm[(size_t)(k0 + i) * DA + k0 + c] : 0.0;
    }
    __syncthreads();
    if (w == 0) {
        // ---- the diagonal block, one wave, rows in registers (k_fit_wide's step 1)
        const int row = lane < nb ? lane : nb - 1;
        double W[NB];
        double rdg = 1.0;
#pragma unroll
        for (int k = 0; k < NB; ++k) W[k] = Pn[row * PS + k];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j < nb) {     // (wave-uniform)
                const double dj = readlane_f64(W[j], j), rs = rsqrt(dj), piv = dj * rs;
                W[j] = (lane == j) ? piv : W[j] * rs;
                rdg = (lane == j) ? rs : rdg;
#pragma unroll
                for (int k = j + 1; k < NB; ++k) W[k] -= W[j] * readlane_f64(W[j], k);
            }
        }
        if (lane < nb) {
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (k <= lane) Pn[lane * PS + k] = W[k];
            rdp[lane] = rdg;
        }
    }
    __syncthreads();
    // ---- the rows below: x L^T = a, one row per thread (k_fit_wide's step 2)
    for (int rr = tid; rr < nr - nb; rr += NT) {
        double* xr = Pn + (size_t)(nb + rr) * PS;
        double x[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) x[c] = xr[c];
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            if (c < nb) {
                x[c] *= rdp[c];
#pragma unroll
                for (int c2 = c + 1; c2 < NB; ++c2) x[c2] -= x[c] * Pn[c2 * PS + c];
            }
        }
#pragma unroll
        for (int c = 0; c < NB; ++c) xr[c] = (c < nb) ? x[c] : 0.0;
    }
    __syncthreads();
    // the factored panel back: lower part INCLUDING the diagonal, and the right-hand-side row
    for (int e = tid; e < nr * NB; e += NT) {
        const int i = e >> LOG_NB, c = e & (NB - 1);
        if (c < nb && i >= c) Wm[(size_t)(k0 + i) * DA + k0 + c] = Pn[i * PS + c];
    }
}

template <int NB>
__global__ void __launch_bounds__(FITW_NT) k_fitw_update(SampleArgs a, double* scratch, int k0) {
    constexpr int PS = NB + 1, LOG_NB = NB == 32 ? 5 : 4;
    PROMP_SMEM_DECL;
    const int D = a.D, DA = D + 1, NT = FITW_NT;
    const int tid = threadIdx.x, task = blockIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
    const int i16 = lane & 15, kk = lane >> 4;
    double* Pn = (double*)PROMP_SMEM_PTR;             // [tr + 16][PS]: the factored panel's rows below its diagonal block
    double* Wm = scratch + (size_t)task * 2 * DA * DA + (size_t)DA * DA;
    const int nb = (D - k0) < NB ? (D - k0) : NB;
    const int k1 = k0 + nb, tr = DA - k1, tc = D - k1;
    for (int e = tid; e < (tr + 16) * NB; e += NT) {
        const int i = e >> LOG_NB, c = e & (NB - 1);
        Pn[i * PS + c] = (c < nb && i < tr) ? Wm[(size_t)(k1 + i) * DA + k0 + c] : 0.0;
    }
    __syncthreads();
    // rows k1..D, columns k1..D-1, the 16 x 16 tiles of the lower triangle dealt to all waves of the task's workgroups
    const int nti = (tr + 15) >> 4, ntj = (tc + 15) >> 4, ntile = nti * ntj;
    const int nw = (FITW_NT / 64) * (int)gridDim.y;
    for (int tix = w + (FITW_NT / 64) * (int)blockIdx.y; tix < ntile; tix += nw) {
        const int bi = tix / ntj, bj = tix - bi * ntj;
        if (bj > bi) continue;                     // wave-uniform
        const double* pa = Pn + (size_t)(16 * bi + i16) * PS + kk;
        const double* pb = Pn + (size_t)(16 * bj + i16) * PS + kk;
        double* dst[4];
        double old[4];
        bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int io = 16 * bi + kk + 4 * r, jo = 16 * bj + i16;
            ok[r] = io < tr && jo < tc && jo <= io;
            dst[r] = Wm + (size_t)(k1 + (ok[r] ? io : 0)) * DA + k1 + (ok[r] ? jo : 0);
            old[r] = *dst[r];
        }
        f64x4 acc = zero4d();
#pragma unroll
        for (int sidx = 0; sidx < NB / 4; ++sidx) acc = mfma16d(pa[4 * sidx], pb[4 * sidx], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (ok[r]) *dst[r] = old[r] - acc[r];
    }
}

template <int NB>
__global__ void __launch_bounds__(FITW_NT) k_fitw_back(SampleArgs a, double* scratch, int* bad) {
    PROMP_SMEM_DECL;
    const int D = a.D, DA = D + 1, NT = FITW_NT;
    const int tid = threadIdx.x, task = blockIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
    double* yv = (double*)PROMP_SMEM_PTR;
    double* wv = yv + DA;
    double* rdv = wv + DA;                    // 1 / diag(L)
    double* tri = rdv + DA;                   // two triangles of the factor's diagonal blocks
    int* flag = (int*)(tri + 2 * NB * (NB + 1));
    const double* Wm = scratch + (size_t)task * 2 * DA * DA + (size_t)DA * DA;
    // ---- back substitution L^T w = z, z = row D of the factor, NB rows of L at a time (k_fit_wide's)
    for (int e = tid; e < D; e += NT) {
        yv[e] = Wm[(size_t)D * DA + e];
        rdv[e] = 1.0 / Wm[(size_t)e * DA + e];
    }
    if (tid == 0) *flag = 0;
    fitw_back_substitute<NB>(Wm, D, DA, yv, wv, rdv, tri, tid, lane, w);
    for (int e = tid; e < D; e += NT)
        if (wv[e] != wv[e]) *flag = 1;
    __syncthreads();
    const int isbad = *flag;
    if (tid == 0) bad[task] = isbad;
    if (!isbad)
        for (int e = tid; e < D; e += NT) a.coeffs[(long long)task * a.coeff_stride + e] = wv[e];
}
