// promp_kernels_rollout.h -- rollout-side kernels.
//
//   k_policy_step   : one environment step of EVERY environment of the meta-batch: mean network under each task's current
//                     parameters, Gaussian exploration noise drawn on the device (Philox4x32-10 + Box-Muller), action =
//                     mean + exp(log_std) * noise; observation, action and mean are written straight into the sampling
//                     step's slab at row (task, env, t) and only the actions go back to the host
//                     (policies/meta_gaussian_mlp_policy.py:99-157 + samplers/meta_sampler.py:87-125 of the reference:
//                     one sess.run plus a Python loop per environment step).
//   k_point_rollout : a whole fixed-horizon rollout of the 2-D point-mass meta-environment of BASELINE config 1
//                     (run_scripts/pro-mp_run_point_mass.py: normalize(MetaPointEnvCorner())), one thread per environment:
//                       envs/point_envs/point_env_2d_corner.py:37   state += clip(action, -0.2, 0.2)
//                       envs/point_envs/point_env_2d_corner.py:62-81 reward: dense -|s' - g|, dense_squared -|s' - g|^2,
//                                                                     sparse (progress towards the goal once the point has
//                                                                     left the start region and the goal is the nearest corner)
//                       envs/normalized_env.py:109-123              the wrapper maps policy actions in [-10, 10] onto the
//                                                                     environment's action box [-0.2, 0.2] and clips
//                     no early termination (point_env_2d_corner.py:39).  The exploration noise is either given by the caller
//                     (reproducible from a host RNG; the parity tests compare with the reference environment step by step)
//                     or drawn on the device.
#pragma once
#include "promp_device.h"

// ---- Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter-based, so every
// (environment, time step) draws its numbers independently of launch geometry.  oracle/philox.py restates it and is
// pinned by the published known-answer vectors.
struct Philox4 {
    unsigned x, y, z, w;
};
PROMP_HD Philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 r;
    r.x = c0; r.y = c1; r.z = c2; r.w = c3;
    return r;
}
// two standard normals from two 32-bit words (Box-Muller; u1 in (0, 1], u2 in [0, 1))
PROMP_HD void box_muller(unsigned a, unsigned b, float& n0, float& n1) {
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u1));
    const float ang = 6.283185307179586f * u2;
    n0 = r * cosf(ang);
    n1 = r * sinf(ang);
}
// action-noise of row `row`, action pair `pair` (actions 2 pair, 2 pair + 1), stream `stream` (sampling step)
PROMP_HD void action_noise(unsigned long long seed, unsigned long long row, unsigned pair, unsigned stream, float& n0, float& n1) {
    const Philox4 r = philox4x32_10((unsigned)row, (unsigned)(row >> 32), pair, stream, (unsigned)seed, (unsigned)(seed >> 32));
    box_muller(r.x, r.y, n0, n1);
}

// mean network of one observation row under flat parameters th (hidden widths <= 128, act_dim <= 8)
PROMP_DEV void mlp_mean(const float* th, const float* x, int O, int A, int H1, int H2, float* mean) {
    const int ob1 = O * H1, oW2 = ob1 + H1, ob2 = oW2 + H1 * H2, oW3 = ob2 + H2, ob3 = oW3 + H2 * A;
    float h1[128], h2[128];
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
        mean[j] = z;
    }
}

struct PolicyStepArgs {
    const float* obs_in;        // [tasks][B][O] observations of this environment step
    const float* theta_tasks;   // [tasks][Theta]
    float *obs, *act, *mean;    // where the rows go: row = env * row_env_stride + t * row_t_stride, env = task * B + b
    long long row_env_stride, row_t_stride;   // fixed-length rollout: (T, 1) into the slab; ragged collection: (1, tasks * B) into the staging rows
    float* old_ls;              // [tasks][A] log_std reported in agent_infos (written at t = 0)
    float* actions_out;         // [tasks][B][A]
    int B, T, t, O, A, H1, H2, NP;
    int clip_infos;             // pre-update policy: agent_infos carry max(log_std, log(min_std)); the noise scale never does
    float min_log_std;
    unsigned long long seed;
    unsigned stream;
};

// grid = (ceil(B / 64), tasks), block = 64: one thread per environment
__global__ void __launch_bounds__(64) k_policy_step(PolicyStepArgs a) {
    const int task = blockIdx.y, b = blockIdx.x * 64 + threadIdx.x;
    const float* th = a.theta_tasks + (long long)task * a.NP;
    const int oS = a.NP - a.A;
    if (a.t == 0 && blockIdx.x == 0 && (int)threadIdx.x < a.A) {
        const float ls = th[oS + threadIdx.x];
        a.old_ls[task * a.A + threadIdx.x] = a.clip_infos ? fmaxf(ls, a.min_log_std) : ls;
    }
    if (b >= a.B) return;
    const long long env = (long long)task * a.B + b, row = env * a.row_env_stride + a.t * a.row_t_stride;
    const float* x = a.obs_in + env * a.O;
    float mean[8];
    mlp_mean(th, x, a.O, a.A, a.H1, a.H2, mean);
    for (int k = 0; k < a.O; ++k) a.obs[row * a.O + k] = x[k];
    for (int j = 0; j < a.A; j += 2) {
        float n0, n1;
        action_noise(a.seed, (unsigned long long)row, (unsigned)(j >> 1), a.stream, n0, n1);
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

// Ragged collection (samplers/meta_sampler.py:100-125: an environment that reports `done` starts its next episode at once):
// promp_policy_step files the rows of vectorised step s under (s, env) in a staging area; when the host knows the episodes, the
// finished ones are copied into the step's slab in path order: path p = staging rows (start[p] + t, env[p]), t < len[p].
struct GatherPathsArgs {
    const float *obs_in, *act_in, *mean_in;   // staging rows [steps][n_envs]
    float *obs, *act, *mean;                  // the slab
    const int *path_env, *path_start, *path_row_offsets;
    int n_envs, O, A;
};
// grid = paths, block = 256
__global__ void __launch_bounds__(256) k_gather_paths(GatherPathsArgs a) {
    const int p = blockIdx.x, env = a.path_env[p], s0 = a.path_start[p], r0 = a.path_row_offsets[p], n = a.path_row_offsets[p + 1] - r0;
    const int W = a.O + 2 * a.A;
    for (int e = threadIdx.x; e < n * W; e += 256) {
        const int t = e / W, k = e - t * W;
        const long long src = (long long)(s0 + t) * a.n_envs + env, dst = r0 + t;
        if (k < a.O) a.obs[dst * a.O + k] = a.obs_in[src * a.O + k];
        else if (k < a.O + a.A) a.act[dst * a.A + (k - a.O)] = a.act_in[src * a.A + (k - a.O)];
        else a.mean[dst * a.A + (k - a.O - a.A)] = a.mean_in[src * a.A + (k - a.O - a.A)];
    }
}

enum { POINT_REWARD_DENSE = 0, POINT_REWARD_DENSE_SQUARED = 1, POINT_REWARD_SPARSE = 2 };

// |p - c|_2 the way a row-wise 2-norm rounds it (squares, sum, root; no fused multiply-add): the sparse reward compares the
// goal's distance with the corners' distances for EQUALITY, so every distance must come out of the same arithmetic
PROMP_HD double point_distance(double p0, double p1, double c0, double c1) {
#pragma clang fp contract(off)
    const double d0 = p0 - c0, d1 = p1 - c1;
    const double q0 = d0 * d0, q1 = d1 * d1;
    return sqrt(q0 + q1);
}

struct PointRolloutArgs {
    const float* theta_tasks;   // [tasks][Theta]
    int NP, H1, H2;
    int B, T;                   // environments per task, horizon
    const double* goals;        // [tasks][2]
    const double* start;        // [tasks][B][2] initial states
    const float* noise;         // [tasks][B][T][2] standard normals, or NULL: drawn on the device from (seed, stream)
    unsigned long long seed;
    unsigned stream;
    float *obs, *act, *rew, *mean;   // slab rows ((task * B + b) * T + t)
    float* old_ls;              // [tasks][2] log_std reported in agent_infos
    int clip_infos;
    float min_log_std;
    double normalization_scale; // the normalize wrapper's policy-side action box half-width (10); 0: bare environment
    double max_step;            // the environment's action box [-max_step, max_step] (0.2)
    int reward_type;
    double sparse_radius;       // 0.5
};

// one step of the normalised point environment: the state advances in place, the reward comes back
PROMP_DEV double point_env_step(const PointRolloutArgs& a, double& s0, double& s1, double g0, double g1, float a0, float a1) {
    // normalize wrapper: lb + (a + s) (ub - lb) / (2 s), clipped to the action box; then the environment's own clip
    // (same bounds).  Same operation order as normalized_env.py:113-114 so that float64 states agree bit for bit.
    const double lb = -a.max_step, ub = a.max_step, sc = a.normalization_scale;
    const double e0 = sc > 0.0 ? lb + ((double)a0 + sc) * (ub - lb) / (2.0 * sc) : (double)a0;
    const double e1 = sc > 0.0 ? lb + ((double)a1 + sc) * (ub - lb) / (2.0 * sc) : (double)a1;
    const double d0 = fmin(fmax(e0, lb), ub), d1 = fmin(fmax(e1, lb), ub);
    const double p0 = s0, p1 = s1;
    s0 += d0;
    s1 += d1;
    const double dist = point_distance(s0, s1, g0, g1);
    double r;
    if (a.reward_type == POINT_REWARD_DENSE) {
        r = -dist;
    } else if (a.reward_type == POINT_REWARD_DENSE_SQUARED) {
        r = -(dist * dist);
    } else {
        r = 0.0;
        if (fabs(s0) + fabs(s1) >= a.sparse_radius) {        // left the start region (L1 norm)
            double nearest = dist;
            for (int c = 0; c < 4; ++c) {                    // corners (-2,-2), (2,-2), (-2,2), (2,2)
                const double c0 = (c & 1) ? 2.0 : -2.0, c1 = (c & 2) ? 2.0 : -2.0;
                nearest = fmin(nearest, point_distance(s0, s1, c0, c1));
            }
            if (dist == nearest) r = point_distance(p0, p1, g0, g1) - dist;   // progress
        }
    }
    return r;
}

// grid = tasks, block = 64
__global__ void __launch_bounds__(64) k_point_rollout(PointRolloutArgs a) {
    const int task = blockIdx.x;
    const float* th = a.theta_tasks + (long long)task * a.NP;
    const int oS = a.NP - 2;
    const float ls0 = th[oS], ls1 = th[oS + 1];
    if (threadIdx.x == 0) {
        a.old_ls[task * 2 + 0] = a.clip_infos ? fmaxf(ls0, a.min_log_std) : ls0;
        a.old_ls[task * 2 + 1] = a.clip_infos ? fmaxf(ls1, a.min_log_std) : ls1;
    }
    const float sd0 = expf(ls0), sd1 = expf(ls1);
    const double g0 = a.goals[task * 2], g1 = a.goals[task * 2 + 1];
    for (int b = threadIdx.x; b < a.B; b += 64) {
        const long long env = (long long)task * a.B + b;
        double s0 = a.start[env * 2], s1 = a.start[env * 2 + 1];
        for (int t = 0; t < a.T; ++t) {
            const long long row = env * a.T + t;
            const float o[2] = {(float)s0, (float)s1};
            float m[2], n0, n1;
            mlp_mean(th, o, 2, 2, a.H1, a.H2, m);
            if (a.noise != nullptr) {
                n0 = a.noise[row * 2];
                n1 = a.noise[row * 2 + 1];
            } else {
                action_noise(a.seed, (unsigned long long)row, 0u, a.stream, n0, n1);
            }
            const float a0 = fmaf(sd0, n0, m[0]), a1 = fmaf(sd1, n1, m[1]);
            a.obs[row * 2] = o[0];  a.obs[row * 2 + 1] = o[1];
            a.mean[row * 2] = m[0];  a.mean[row * 2 + 1] = m[1];
            a.act[row * 2] = a0;  a.act[row * 2 + 1] = a1;
            const double r = point_env_step(a, s0, s1, g0, g1, a0, a1);
            a.rew[row] = (float)r;
        }
    }
}
