// promp_kernels_rollout.h -- a whole rollout on the device for environments whose physics is a few lines of
// arithmetic: the 2-D point-mass meta-environment of BASELINE config 0
// (run_scripts/pro-mp_run_point_mass.py; the reference steps it in NumPy, envs/point_envs/point_env_2d_corner.py).
//
// Host-side rollouts pay one policy query (upload observations, launch, download means) plus a Python loop over
// environments PER ENVIRONMENT STEP.  Here one thread owns one environment for the whole horizon: policy forward
// (under its task's current parameters), Gaussian action, state update and reward, written straight into the
// [task x path x t] slab that promp_process_samples reads -- no per-step round trip and no upload afterwards.
// The exploration noise is an input ([tasks][envs][T][2] standard normals drawn by the caller), which keeps the
// trajectory reproducible from a host RNG and comparable with the NumPy environment step by step.
#pragma once
#include "promp_device.h"

struct PointRolloutArgs {
    const float* theta_tasks;   // [tasks][Theta]
    int NP, H1, H2;
    int B, T;                   // environments per task, horizon
    const double* goals;        // [tasks][2]
    const double* start;        // [tasks][B][2] initial states
    const float* noise;         // [tasks][B][T][2]
    float *obs, *act, *rew, *mean;   // slab rows ((task * B + b) * T + t)
    float* old_ls;              // [tasks][2] log_std reported in agent_infos
    int clip_infos;             // pre-update policy: agent_infos carry max(log_std, log(min_std)); the noise scale never does
    float min_log_std;
    double max_step;            // per-coordinate action clip of the environment (0.1)
};

// grid = tasks, block = 64
__global__ void __launch_bounds__(64) k_point_rollout(PointRolloutArgs a) {
    const int task = blockIdx.x;
    const int H1 = a.H1, H2 = a.H2;
    const int ob1 = 2 * H1, oW2 = ob1 + H1, ob2 = oW2 + H1 * H2, oW3 = ob2 + H2, ob3 = oW3 + H2 * 2, oS = ob3 + 2;
    const float* th = a.theta_tasks + (long long)task * a.NP;
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
            const float o0 = (float)s0, o1 = (float)s1;
            float h1[128], h2[128];
            for (int j = 0; j < H1; ++j) h1[j] = fast_tanh(fmaf(o1, th[H1 + j], fmaf(o0, th[j], th[ob1 + j])));
            for (int j = 0; j < H2; ++j) {
                float z = th[ob2 + j];
                for (int k = 0; k < H1; ++k) z = fmaf(h1[k], th[oW2 + k * H2 + j], z);
                h2[j] = fast_tanh(z);
            }
            float m0 = th[ob3], m1 = th[ob3 + 1];
            for (int k = 0; k < H2; ++k) {
                m0 = fmaf(h2[k], th[oW3 + k * 2], m0);
                m1 = fmaf(h2[k], th[oW3 + k * 2 + 1], m1);
            }
            const float a0 = fmaf(sd0, a.noise[row * 2], m0), a1 = fmaf(sd1, a.noise[row * 2 + 1], m1);
            a.obs[row * 2] = o0;  a.obs[row * 2 + 1] = o1;
            a.mean[row * 2] = m0;  a.mean[row * 2 + 1] = m1;
            a.act[row * 2] = a0;  a.act[row * 2 + 1] = a1;
            const double d0 = fmin(fmax((double)a0, -a.max_step), a.max_step), d1 = fmin(fmax((double)a1, -a.max_step), a.max_step);
            s0 += d0;
            s1 += d1;
            a.rew[row] = (float)(-sqrt((s0 - g0) * (s0 - g0) + (s1 - g1) * (s1 - g1)));
        }
    }
}
