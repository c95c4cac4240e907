"""Host-side rollout collection: environment pools and MetaSampler (contract of reference samplers/meta_sampler.py:59-137
and samplers/vectorized_env_executor.py:7-177; the reference's own checks are tests/test_samplers.py:132-189)."""
import numpy as np

from promp_amd.samplers.meta_sampler import MetaSampler
from promp_amd.samplers.vectorized_env_executor import MetaIterativeEnvExecutor, MetaParallelEnvExecutor


class CounterEnv(object):
    """deterministic toy: obs = [task, episode index, t]; terminates by itself after `task + 2` steps when task is odd"""

    def __init__(self):
        self.task, self.episode, self.t = 0, -1, 0

    def sample_tasks(self, n):
        return list(range(n))

    def set_task(self, task):
        self.task = task

    def reset(self):
        self.episode += 1
        self.t = 0
        return np.array([self.task, self.episode, self.t], dtype=np.float32)

    def step(self, action):
        self.t += 1
        done = bool(self.task % 2 == 1 and self.t >= self.task + 2)
        info = dict(t=self.t, nested=dict(twice=2.0 * self.t))
        return np.array([self.task, self.episode, self.t], dtype=np.float32), float(np.sum(action)) + self.t, done, info


class EchoPolicy(object):
    """actions = [obs[2], task index]; agent_infos carry mean / log_std like the Gaussian policy"""

    def __init__(self, M):
        self.M = M

    def get_actions(self, observations):
        assert len(observations) == self.M
        actions = [np.stack([o[:, 2], np.full(len(o), i)], axis=1).astype(np.float32) for i, o in enumerate(observations)]
        infos = [[dict(mean=a, log_std=np.zeros(2, np.float32)) for a in acts] for acts in actions]
        return actions, infos


def _drive(pool, n_steps, M, per):
    pool.set_tasks(list(range(M)))
    trace = [np.asarray(pool.reset())]
    for s in range(n_steps):
        obs, rew, done, infos = pool.step([np.array([s, e], np.float32) for e in range(M * per)])
        trace.append((np.asarray(obs), np.asarray(rew), np.asarray(done), [i['nested']['twice'] for i in infos]))
    return trace


def test_iterative_pool_resets_on_done_and_on_the_horizon():
    M, per, T = 3, 2, 4
    pool = MetaIterativeEnvExecutor(CounterEnv(), M, per, T)
    assert pool.num_envs == M * per
    tr = _drive(pool, 9, M, per)
    np.testing.assert_array_equal(tr[0][:, 0], [0, 0, 1, 1, 2, 2])            # task k -> environments [k per, (k+1) per)
    for s, (obs, rew, done, _) in enumerate(tr[1:], start=1):
        # even tasks never finish by themselves: the horizon (T = 4) ends their episodes; task 1 finishes after 3 steps
        np.testing.assert_array_equal(done[[0, 1, 4, 5]], [s % T == 0] * 4)
        np.testing.assert_array_equal(done[[2, 3]], [s % 3 == 0] * 2)
        # a finished environment hands back the first observation of its next episode
        assert all(obs[e, 2] == 0 for e in np.flatnonzero(done))
        assert all(obs[e, 2] > 0 for e in np.flatnonzero(~done))


def test_parallel_pool_matches_iterative_pool():
    M, per, T = 3, 2, 4
    a = _drive(MetaIterativeEnvExecutor(CounterEnv(), M, per, T), 7, M, per)
    pool = MetaParallelEnvExecutor(CounterEnv(), M, per, T)
    try:
        b = _drive(pool, 7, M, per)
    finally:
        pool.close()
    np.testing.assert_array_equal(a[0], b[0])
    for x, y in zip(a[1:], b[1:]):
        for u, v in zip(x, y):
            np.testing.assert_array_equal(np.asarray(u), np.asarray(v))


def test_meta_sampler_files_finished_trajectories_per_task():
    M, P, T = 4, 3, 5
    sampler = MetaSampler(CounterEnv(), EchoPolicy(M), rollouts_per_meta_task=P, meta_batch_size=M, max_path_length=T)
    sampler.update_tasks()
    paths = sampler.obtain_samples()
    assert list(paths.keys()) == list(range(M))
    total = 0
    for task, plist in paths.items():
        assert len(plist) >= P
        for p in plist:
            assert set(p.keys()) == {'observations', 'actions', 'rewards', 'env_infos', 'agent_infos'}
            n = len(p['rewards'])
            total += n
            assert n == (task + 2 if task % 2 == 1 and task + 2 < T else T)            # own termination or the horizon
            assert p['observations'].shape == (n, 3) and p['actions'].shape == (n, 2)
            np.testing.assert_array_equal(p['observations'][:, 0], task)               # filed under its own task
            np.testing.assert_array_equal(p['observations'][:, 2], np.arange(n))       # one episode, in order
            np.testing.assert_array_equal(p['actions'][:, 0], np.arange(n))
            np.testing.assert_array_equal(p['env_infos']['t'], np.arange(1, n + 1))
            np.testing.assert_array_equal(p['env_infos']['nested']['twice'], 2.0 * np.arange(1, n + 1))
            np.testing.assert_array_equal(p['agent_infos']['mean'], p['actions'])
            assert p['agent_infos']['log_std'].shape == (n, 2)
            np.testing.assert_allclose(p['rewards'], p['actions'].sum(axis=1) + np.arange(1, n + 1))
    assert total >= sampler.total_samples == M * P * T
    assert sampler.total_timesteps_sampled == M * P * T
