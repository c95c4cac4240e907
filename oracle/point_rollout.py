"""Oracle for the device rollout of the 2-D point-mass meta-environment (TEST INFRASTRUCTURE ONLY).

Restates, in float64 NumPy, the environment BASELINE config 1 trains on, run_scripts/pro-mp_run_point_mass.py:
normalize(MetaPointEnvCorner()):

  * envs/normalized_env.py:109-123        the wrapper rescales a policy action a in [-10, 10] to the environment's action box
                                          [lb, ub] = [-0.2, 0.2]: lb + (a + 10) (ub - lb) / 20 = 0.02 a, then clips to the box
  * envs/point_envs/point_env_2d_corner.py:37     state <- state + clip(action, -0.2, 0.2)
  * envs/point_envs/point_env_2d_corner.py:39     done = False (fixed horizon)
  * envs/point_envs/point_env_2d_corner.py:62-81  reward: 'dense' -|s' - g|_2; 'dense_squared' -|s' - g|_2^2; 'sparse': 0 while
                                          |s'|_1 < sparse_reward_radius (0.5); else, if the goal is the nearest of the four corners
                                          (+-2, +-2), the progress |s - g|_2 - |s' - g|_2, else 0
  * envs/point_envs/point_env_2d_corner.py:48     reset: state ~ U(-0.2, 0.2)^2 (drawn by the caller here)

`env_step` is pinned by tests/golden/point_env_*.npz (outputs of the reference classes, oracle/gen_golden.py).
`rollout` adds the policy of policies/meta_gaussian_mlp_policy.py:99-157 (action = mean + noise * exp(log_std)) with the
exploration noise given, and returns the flattened [task x env x t] arrays the device writes."""
import numpy as np

from . import policy as op

CORNERS = np.array([[-2.0, -2.0], [2.0, -2.0], [-2.0, 2.0], [2.0, 2.0]])


def env_step(state, action, goal, reward_type='sparse', normalization_scale=10.0, max_step=0.2, sparse_radius=0.5):
    """one step of normalize(MetaPointEnvCorner) for a batch: state [B,2], action [B,2] (policy scale) -> (next state, reward);
    normalization_scale = 0: the bare environment"""
    state = np.asarray(state, dtype=np.float64)
    action = np.asarray(action, dtype=np.float64)
    lb, ub, s = -max_step, max_step, normalization_scale
    scaled = lb + (action + s) * (ub - lb) / (2 * s) if s > 0 else action       # normalized_env.py:113
    move = np.clip(np.clip(scaled, lb, ub), -max_step, max_step)                # :114, then point_env_2d_corner.py:37
    nxt = state + move
    dist = np.linalg.norm(nxt - goal, axis=1)
    if reward_type == 'dense':
        return nxt, -dist
    if reward_type == 'dense_squared':
        return nxt, -dist ** 2
    assert reward_type == 'sparse'
    corner_dist = np.stack([np.linalg.norm(nxt - c, axis=1) for c in CORNERS])
    progress = np.linalg.norm(state - goal, axis=1) - dist
    outside = np.sum(np.abs(nxt), axis=1) >= sparse_radius
    towards_nearest = dist == np.minimum(dist, corner_dist.min(axis=0))
    return nxt, np.where(outside & towards_nearest, progress, 0.0)


def rollout(spec, theta_tasks, goals, start, noise, clip_infos=True, **env):
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
            state, rew[i, :, t] = env_step(state, a.astype(np.float32).astype(np.float64), np.asarray(goals[i], np.float64), **env)
    return dict(obs=obs.reshape(-1, 2), act=act.reshape(-1, 2), mean=mean.reshape(-1, 2), rew=rew.reshape(-1),
                log_std=log_std)
