// promp_kernels_policy.h -- task-level reductions, Adam and the rollout-time forward pass (reference rows a10-a14).
//
//   k_reduce_task  : fixed-order sum of a task's per-workgroup partials + the update that consumes it (the cooperative
//                    kernels of promp_kernels_policy_wide.h; the chain kernels do this inside the pass launch)
//   k_reduce_final : task sum of the per-task gradients and scalars                          (K14)
//   k_mean_adam    : mean over the global meta-batch + tf.train.AdamOptimizer step            (K14)
//   k_final_adam   : the two above in one launch (single rank)
//   k_policy_forward : mean network under each task's current parameters (get_actions)
#pragma once
#include "promp_device.h"
#include "promp_kernels_chain.h"
#include "promp_kernels_pass.h"

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
    // mode 0 only, optional: the same step / scalars a second time (promp_inner_adapt leaves its result where the first
    // epoch of the following optimisation would have computed it: its adaptation chain and inner scalars)
    float* next2;
    float* scal2;
};

__global__ void __launch_bounds__(256) k_reduce_task(ReduceArgs a) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int task = blockIdx.y;
    if (j >= a.NP + 2) return;
    if (a.mode == 4 && j < a.NP) return;
    float g = 0.f;
    const int wg0 = a.task_wg_offsets[task], wg1 = a.task_wg_offsets[task + 1];
    // eight rows per round, all requested together (the rows sit in other CUs' cache lines or in memory: the kernel is one
    // or two memory round trips long, so what matters is how many loads are in flight); rows past the task's last are
    // clamped to a valid row and masked by a multiplication, which keeps the loads unconditional.  Added in slot order.
    for (int wb = wg0; wb < wg1; wb += 8) {
        float x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int wg = wb + q < wg1 ? wb + q : wg1 - 1;
            x[q] = a.partials[(long long)wg * a.partial_stride + j] * (wb + q < wg1 ? 1.f : 0.f);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) g += x[q];
    }
    if (j >= a.NP) {
        a.scal[task * 2 + (j - a.NP)] = g;
        if (a.scal2 != nullptr) a.scal2[task * 2 + (j - a.NP)] = g;
        return;
    }
    const long long tj = (long long)task * a.NP + j;
    if (a.mode == 0) {
        const float nx = a.cur[(long long)task * a.cur_task_stride + j] - a.step_sizes[j] * g;
        a.next[tj] = nx;
        if (a.next2 != nullptr) a.next2[tj] = nx;
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
            // (up to 12 tasks of the quarter requested together, clamped + masked like k_reduce_task; order i = q, q + 4, ...)
            for (int ib = q; ib < a.n_tasks; ib += 48) {
                float x[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int i = ib + 4 * u;
                    x[u] = a.lam[(long long)(i < a.n_tasks ? i : q) * a.NP + j] * (i < a.n_tasks ? 1.f : 0.f);
                }
#pragma unroll
                for (int u = 0; u < 12; ++u) s += x[u];
            }
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
#define PROMP_ETA_MAX 8        // most inner gradient steps a context is created for
struct AdamArgs {
    float* theta;
    float* m;
    float* v;
    const float* red;
    float* grad_mean;  // [NP] task-mean gradient (kept for promp_meta_grad's output)
    float* stats;      // [K+2]
    float eta[PROMP_ETA_MAX];   // inner KL coefficients, by value (K <= PROMP_ETA_MAX): no host -> device copy per optimisation
    int NP, K, A;
    float inv_tasks;
    float lr_t;        // lr * sqrt(1-b2^t)/(1-b1^t); 0 => no parameter update (stats / grad only)
    int do_update;
    int n_trainable;   // parameters [n_trainable, NP) are left alone (learn_std = False: the trailing log_std entries)
    // promp_optimize_begin's last launch: both statistics slots go straight to page-locked host memory, followed by a
    // sequence number the host polls (no copy operation, no event on the queue)
    float* host_stats;       // [2 (K + 2) + 1] or NULL; the last entry: the smallest log_std entry of theta as it is now
    unsigned* host_seq;
    unsigned seq;
};
// (called by the one thread that has just written stats[0 .. K+2); the second slot was written by an earlier launch)
PROMP_DEV void publish_stats(const AdamArgs& a) {
    if (a.host_stats == nullptr) return;
    for (int i = 0; i < 2 * (a.K + 2); ++i) a.host_stats[i] = a.stats[i];
    {   // (the publishing launch does not update theta: what is read here is what the next inner step will see)
        float lo = a.theta[a.NP - a.A];
        for (int i = 1; i < a.A; ++i) lo = fminf(lo, a.theta[a.NP - a.A + i]);
        a.host_stats[2 * (a.K + 2)] = lo;
    }
    release_store_system(a.host_seq, a.seq);
}

__global__ void __launch_bounds__(256) k_mean_adam(AdamArgs a) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < a.NP) {
        const float g = a.red[j] * a.inv_tasks;
        a.grad_mean[j] = g;
        if (a.do_update && j < a.n_trainable) {
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
        publish_stats(a);
    }
}

// Fixed-order exchange: the ranks' vectors side by side ([nranks][n], ncclAllGather) added in rank order -- the same additions in
// the same order on every rank.
__global__ void __launch_bounds__(256) k_sum_ranks(const float* gathered, float* out, int n, int nranks) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    float s = gathered[j];
    for (int r = 1; r < nranks; ++r) s += gathered[(long long)r * n + j];
    out[j] = s;
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
            // (up to 12 tasks of the quarter requested together, clamped + masked like k_reduce_task; order i = q, q + 4, ...)
            for (int ib = q; ib < a.n_tasks; ib += 48) {
                float x[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int i = ib + 4 * u;
                    x[u] = a.lam[(long long)(i < a.n_tasks ? i : q) * a.NP + j] * (i < a.n_tasks ? 1.f : 0.f);
                }
#pragma unroll
                for (int u = 0; u < 12; ++u) s += x[u];
            }
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
            if (ad.do_update && j < ad.n_trainable) {
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
        publish_stats(ad);
    }
}

// dst[i][:] = src[:] for i < n_tasks   (MetaPolicy.switch_to_pre_update)
__global__ void __launch_bounds__(256) k_replicate(float* dst, const float* src, int NP, int n_tasks) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < NP)
        for (int i = 0; i < n_tasks; ++i) dst[(long long)i * NP + j] = src[j];
}

// Elementwise glue of promp_constraint_hvp (products with the adaptation Jacobian J = I - diag(alpha) H):
//   mode 0:  dir += alpha * lam                       (dir <- J dir, lam = -H dir from the pass before)
//   mode 1:  w = -lam ;      dir = alpha * w          (w <- H_KL u, lam = -H_KL u)
//   mode 2:  w += lam ;      dir = alpha * w          (w <- J^T w, lam = -H (alpha * w))
// grid = (ceil(NP / 256), tasks)
__global__ void __launch_bounds__(256) k_jstep(float* dir, float* w, const float* lam, const float* alpha, int NP, int mode) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= NP) return;
    const long long tj = (long long)blockIdx.y * NP + j;
    if (mode == 0) {
        dir[tj] += alpha[j] * lam[tj];
        return;
    }
    const float x = (mode == 1) ? -lam[tj] : w[tj] + lam[tj];
    w[tj] = x;
    dir[tj] = alpha[j] * x;
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

// ---------------------------------------------------------------------------------------------
// Conjugate gradients on the device (promp_cg_solve): the vector side of ConjugateGradientOptimizer's solve
// (optimizers/conjugate_gradient_optimizer.py:325-354, conjugate_gradients) between two products, in ONE launch of one workgroup -- the vectors are
// Theta floats (6 k ... 70 k), the launch is a link in a chain of dependent launches and costs what any of them costs.
// Sums in a fixed order (thread-strided, then a tree over the workgroup) in float64; the vectors stay float32 like the
// NumPy arrays of the host form.
// ---------------------------------------------------------------------------------------------
// theta = theta0 + s x  (FiniteDifferenceHvp: the parameters the displaced constraint gradient is taken at; one rounding)
__global__ void __launch_bounds__(256) k_cg_displace(float* theta, const float* theta0, const float* x, float s, int n) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) theta[j] = __builtin_fmaf(s, x[j], theta0[j]);
}

struct CgArgs {
    const float* g_ahead;    // finite differences: the constraint gradient at theta0 + eps v; exact: the task sums of the product
    const float* g_behind;   // ... at theta0 - eps v (symmetric) / at theta0 (one-sided); NULL: nothing is subtracted
    float div_h, mul_s;      // H v = ((g_ahead - g_behind) / div_h) * mul_s   (2 eps or eps, 1; exact: 1, 1 / tasks)
    float reg;               // + reg v
    float *x, *r, *d, *hd;   // solution, residual, search direction, (H + reg I) v
    double* scal;            // [0] r.r   [1] d.(H + reg I)d of the last iteration   [2] != 0: converged   [3] x.(H + reg I)x
    float tol;               // residual_tol
    int n;
    int mode;                // 2: x = 0, d = r (= b), scal[0] = r.r;   0: one iteration along d;   1: the closing product on x
};
PROMP_DEV double cg_block_sum(double v, double* buf) {
    const int t = threadIdx.x;
    __syncthreads();
    buf[t] = v;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (t < s) buf[t] += buf[t + s];
        __syncthreads();
    }
    return buf[0];
}
// grid = 1, block = 1024
__global__ void __launch_bounds__(1024) k_cg_step(CgArgs a) {
    __shared__ double buf[1024];
    const int t = threadIdx.x;
    if (a.mode == 2) {
        double s = 0.0;
        for (int j = t; j < a.n; j += 1024) {
            const float b = a.r[j];
            a.x[j] = 0.f;
            a.d[j] = b;
            s += (double)b * (double)b;
        }
        s = cg_block_sum(s, buf);
        if (t == 0) {
            a.scal[0] = s; a.scal[1] = 0.0; a.scal[2] = 0.0; a.scal[3] = 0.0;
        }
        return;
    }
    const float* v = a.mode == 1 ? a.x : a.d;
    double dot = 0.0;
    for (int j = t; j < a.n; j += 1024) {
        const float gb = a.g_behind ? a.g_behind[j] : 0.f;
        const float h = __builtin_fmaf(a.reg, v[j], ((a.g_ahead[j] - gb) / a.div_h) * a.mul_s);
        a.hd[j] = h;
        dot += (double)v[j] * (double)h;
    }
    dot = cg_block_sum(dot, buf);
    if (a.mode == 1) {
        if (t == 0) a.scal[3] = dot;
        return;
    }
    const double res = a.scal[0];
    if (a.scal[2] != 0.0) return;          // converged in an earlier iteration: the host form has left its loop (uniform: no barrier is skipped by a part of the block)
    const float step = (float)(res / dot);
    double nres = 0.0;
    for (int j = t; j < a.n; j += 1024) {
        a.x[j] = __builtin_fmaf(step, a.d[j], a.x[j]);
        const float rr = __builtin_fmaf(-step, a.hd[j], a.r[j]);
        a.r[j] = rr;
        nres += (double)rr * (double)rr;
    }
    nres = cg_block_sum(nres, buf);
    const float beta = (float)(nres / res);
    for (int j = t; j < a.n; j += 1024) a.d[j] = __builtin_fmaf(beta, a.d[j], a.r[j]);
    if (t == 0) {
        a.scal[0] = nres;
        a.scal[1] = dot;
        if (nres < (double)a.tol) a.scal[2] = 1.0;
    }
}
