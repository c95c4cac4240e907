"""Seeded synthetic trajectory batches of the shape MetaSampler.obtain_samples returns
(reference: meta_policy_search/samplers/meta_sampler.py:59-137), as specified in
SURVEY.md section 8(d) "synthetic inputs".  NumPy only; used by tests and bench.py to feed
the HIP path and the oracle the *same* inputs.
"""
from collections import OrderedDict
import numpy as np

CONFIGS = {
    # BASELINE.json configs[0..3]; P = rollouts_per_meta_task = 20 (run-script default)
    1: dict(M=4, P=20, T=100, O=2, A=2, hidden=(32, 32)),
    2: dict(M=8, P=20, T=200, O=20, A=6, hidden=(64, 64)),
    3: dict(M=40, P=20, T=200, O=20, A=6, hidden=(64, 64)),
    4: dict(M=40, P=20, T=200, O=111, A=8, hidden=(128, 128)),
}


def param_count(O, hidden, A):
    sizes = (O,) + tuple(hidden) + (A,)
    return sum(sizes[i] * sizes[i + 1] + sizes[i + 1] for i in range(len(sizes) - 1)) + A


def init_theta(rng, O, hidden, A, init_std=1.0):
    """Xavier-uniform kernels, zero biases, log_std = log(init_std); flat float32 vector in the
    reference's parameter order (policies/networks/mlp.py:12-13, gaussian_mlp_policy.py:63-69)."""
    sizes = (O,) + tuple(hidden) + (A,)
    parts = []
    for i in range(len(sizes) - 1):
        lim = np.sqrt(6.0 / (sizes[i] + sizes[i + 1]))
        parts += [rng.uniform(-lim, lim, size=(sizes[i], sizes[i + 1])).reshape(-1), np.zeros(sizes[i + 1])]
    parts.append(np.full(A, np.log(init_std)))
    return np.concatenate(parts).astype(np.float32)


def mlp_mean(theta, obs, O, hidden, A):
    """Mean network forward (data generation only): tanh hidden layers, linear output."""
    sizes = (O,) + tuple(hidden) + (A,)
    x = np.asarray(obs, dtype=np.float32)
    off = 0
    for i in range(len(sizes) - 1):
        W = theta[off:off + sizes[i] * sizes[i + 1]].reshape(sizes[i], sizes[i + 1]); off += W.size
        b = theta[off:off + sizes[i + 1]]; off += b.size
        x = x @ W + b
        if i < len(sizes) - 2:
            x = np.tanh(x)
    return x.astype(np.float32)


def make_paths(rng, theta_tasks, M, P, T, O, A, hidden, ragged=False, obs_dtype=np.float32):
    """One sampling step's paths_meta_batch: OrderedDict{task -> [path dict] * P}.

    obs ~ 3*N(0,1) + per-task offset U(-2,2); old_mean = pi_{theta_task}(obs); actions =
    old_mean + exp(log_std)*N(0,1); rewards ~ N(-1,1) + 0.1*obs[:,0].
    theta_tasks: [M, Theta] (or [Theta], shared) flat float32 parameters used to produce agent_infos.
    ragged=True draws path lengths in [T//2, T] (correctness tests only).
    """
    theta_tasks = np.asarray(theta_tasks, dtype=np.float32)
    if theta_tasks.ndim == 1:
        theta_tasks = np.broadcast_to(theta_tasks, (M, theta_tasks.size))
    out = OrderedDict()
    for i in range(M):
        offset = rng.uniform(-2, 2, size=O)
        log_std = theta_tasks[i][-A:]
        paths = []
        for _ in range(P):
            Tp = int(rng.randint(max(T // 2, 1), T + 1)) if ragged else T
            obs = (rng.randn(Tp, O) * 3.0 + offset).astype(obs_dtype)
            mean = mlp_mean(theta_tasks[i], obs, O, hidden, A)
            act = (mean + np.exp(log_std) * rng.randn(Tp, A)).astype(np.float32)
            rew = (rng.randn(Tp) - 1.0 + 0.1 * obs[:, 0]).astype(np.float32)
            paths.append(dict(observations=obs, actions=act, rewards=rew, env_infos={},
                              agent_infos=dict(mean=mean, log_std=np.tile(log_std, (Tp, 1)).astype(np.float32))))
        out[i] = paths
    return out


def make_paths_for_tasks(base_seed, task_ids, theta_tasks, P, T, O, A, hidden):
    """Like make_paths but every task draws from its own RandomState(base_seed*100003 + task_id), so a
    rank holding tasks {i : i % world == rank} generates exactly its shard of the single-GPU batch."""
    theta_tasks = np.asarray(theta_tasks, dtype=np.float32)
    out = OrderedDict()
    for slot, i in enumerate(task_ids):
        rng = np.random.RandomState((base_seed * 100003 + int(i)) % (2 ** 31 - 1))
        th = theta_tasks if theta_tasks.ndim == 1 else theta_tasks[slot]
        out[slot] = make_paths(rng, th, 1, P, T, O, A, hidden)[0]
    return out
