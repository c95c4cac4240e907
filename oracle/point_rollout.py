"""Oracle for the device rollout of the 2-D point-mass meta-environment (TEST INFRASTRUCTURE ONLY).

Restates, in float64 NumPy, what promp_rollout_point_env computes: the environment the reference's config 0 trains on
(run_scripts/pro-mp_run_point_mass.py -> envs/point_envs: state += clip(action, -0.1, 0.1), reward = -|state - goal|,
no early termination) stepped for a fixed horizon under each task's tanh-MLP Gaussian policy
(policies/meta_gaussian_mlp_policy.py:99-157: action = mean + noise * exp(log_std)), with the exploration noise given.
Returns the flattened [task x env x t] arrays the device writes."""
import numpy as np

from . import policy as op


def rollout(spec, theta_tasks, goals, start, noise, clip_infos=True, max_step=0.1):
    M, B, T = noise.shape[0], noise.shape[1], noise.shape[2]
    obs = np.zeros((M, B, T, 2))
    act = np.zeros((M, B, T, 2))
    mean = np.zeros((M, B, T, 2))
    rew = np.zeros((M, B, T))
    log_std = np.zeros((M, 2))
    for i in range(M):
        theta = np.asarray(theta_tasks[i], dtype=np.float64)
        raw = theta[-2:]
        log_std[i] = np.maximum(raw, np.log(1e-6)) if clip_infos else raw
        state = np.asarray(start[i], dtype=np.float64).copy()            # [B, 2]
        for t in range(T):
            o = state.astype(np.float32).astype(np.float64)               # observations are handed over as float32
            m = op.forward(spec, theta, o, False)[0]
            a = m + np.exp(raw) * noise[i, :, t]
            obs[i, :, t], mean[i, :, t], act[i, :, t] = o, m, a
            state = state + np.clip(a, -max_step, max_step)
            rew[i, :, t] = -np.linalg.norm(state - goals[i], axis=1)
    return dict(obs=obs.reshape(-1, 2), act=act.reshape(-1, 2), mean=mean.reshape(-1, 2), rew=rew.reshape(-1),
                log_std=log_std)
