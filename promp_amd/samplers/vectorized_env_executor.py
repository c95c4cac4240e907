"""Environment pools for meta-batch rollouts: `envs_per_task` copies of the environment per task, stepped together.

Public contract (what MetaSampler and user code rely on; the reference offers the same two classes,
meta_policy_search/samplers/vectorized_env_executor.py:7-177):

    pool = MetaIterativeEnvExecutor(env, meta_batch_size, envs_per_task, max_path_length)     # in-process
    pool = MetaParallelEnvExecutor(env, meta_batch_size, envs_per_task, max_path_length)      # one process per task
    pool.set_tasks(tasks)                 # task k -> environments [k * envs_per_task, (k + 1) * envs_per_task)
    obs = pool.reset()                    # list[num_envs]
    obs, rewards, dones, env_infos = pool.step(actions)
    pool.num_envs

An environment that reports `done`, or whose episode reaches max_path_length steps, is reset on the spot: its `done` is
True and the observation returned for it is the first observation of the next episode.

Both pools are built on `_EnvBlock` (a block of environments with per-slot step clocks); the parallel pool runs one block
per task in a worker process and talks to it through a small request / reply protocol.  Host side by design: environment
physics is third-party Python / C and not part of the accelerated path.
"""
import copy
import itertools
import multiprocessing as mp
import pickle

import numpy as np


class _EnvBlock(object):
    """A block of environments plus one step clock per slot; the horizon is folded into `done`."""

    def __init__(self, envs, horizon):
        self.envs = list(envs)
        self.horizon = int(horizon)
        self.clock = np.zeros(len(self.envs), dtype=np.int64)

    def __len__(self):
        return len(self.envs)

    def restart(self, _=None):
        self.clock[:] = 0
        return [env.reset() for env in self.envs]

    def assign(self, task):
        for env in self.envs:
            env.set_task(task)

    def advance(self, actions):
        n = len(self.envs)
        assert len(actions) == n, 'one action per environment (%d given, %d environments)' % (len(actions), n)
        observations, env_infos = [None] * n, [None] * n
        rewards, finished = [0.0] * n, np.zeros(n, dtype=bool)
        for slot in range(n):
            observations[slot], rewards[slot], finished[slot], env_infos[slot] = self.envs[slot].step(actions[slot])
        self.clock += 1
        finished |= self.clock >= self.horizon
        for slot in np.flatnonzero(finished):
            observations[slot] = self.envs[slot].reset()
        self.clock[finished] = 0
        return observations, rewards, finished.tolist(), env_infos


class MetaIterativeEnvExecutor(object):
    """All meta_batch_size * envs_per_task environments live in this process and are stepped one after the other."""

    def __init__(self, env, meta_batch_size, envs_per_task, max_path_length):
        self.meta_batch_size, self.envs_per_task = int(meta_batch_size), int(envs_per_task)
        self._block = _EnvBlock((copy.deepcopy(env) for _ in range(self.meta_batch_size * self.envs_per_task)), max_path_length)

    @property
    def num_envs(self):
        return len(self._block)

    @property
    def envs(self):
        return self._block.envs

    def set_tasks(self, tasks):
        assert len(tasks) == self.meta_batch_size
        for k, task in enumerate(tasks):
            for env in self._block.envs[k * self.envs_per_task:(k + 1) * self.envs_per_task]:
                env.set_task(task)

    def reset(self):
        return self._block.restart()

    def step(self, actions):
        observations, rewards, finished, env_infos = self._block.advance(actions)
        return observations, rewards, np.asarray(finished), env_infos


def _serve_block(link, env_bytes, n_envs, horizon, seed):
    """Worker process: owns one task's environments and answers (request, payload) messages until told to stop."""
    np.random.seed(int(seed))
    block = _EnvBlock((pickle.loads(env_bytes) for _ in range(n_envs)), horizon)
    handlers = {'advance': block.advance, 'restart': block.restart, 'assign': block.assign}
    try:
        for request, payload in iter(link.recv, ('stop', None)):
            link.send(handlers[request](payload))
    except (EOFError, KeyboardInterrupt):
        pass
    finally:
        link.close()


class MetaParallelEnvExecutor(object):
    """One daemon worker process per task; every call scatters one request per worker, then gathers the replies in task
    order (so the workers run concurrently)."""

    def __init__(self, env, meta_batch_size, envs_per_task, max_path_length):
        self.meta_batch_size, self.envs_per_task = int(meta_batch_size), int(envs_per_task)
        env_bytes = pickle.dumps(env)
        # distinct seeds so that the workers' environments do not replay one another's randomness
        seeds = np.random.SeedSequence(int(np.random.randint(0, 2 ** 31 - 1))).generate_state(self.meta_batch_size)
        self._links, self._workers = [], []
        for seed in seeds:
            here, there = mp.Pipe()
            proc = mp.Process(target=_serve_block, args=(there, env_bytes, self.envs_per_task, max_path_length, seed), daemon=True)
            proc.start()
            there.close()
            self._links.append(here)
            self._workers.append(proc)

    @property
    def num_envs(self):
        return self.meta_batch_size * self.envs_per_task

    def _round_trip(self, request, payloads):
        for link, payload in zip(self._links, payloads):
            link.send((request, payload))
        return [link.recv() for link in self._links]

    def set_tasks(self, tasks=None):
        assert len(tasks) == self.meta_batch_size
        self._round_trip('assign', tasks)

    def reset(self):
        return list(itertools.chain.from_iterable(self._round_trip('restart', [None] * self.meta_batch_size)))

    def step(self, actions):
        assert len(actions) == self.num_envs
        per = self.envs_per_task
        replies = self._round_trip('advance', [actions[k * per:(k + 1) * per] for k in range(self.meta_batch_size)])
        observations, rewards, finished, env_infos = (list(itertools.chain.from_iterable(column)) for column in zip(*replies))
        return observations, rewards, finished, env_infos

    def close(self):
        for link in self._links:
            try:
                link.send(('stop', None))
                link.close()
            except (OSError, BrokenPipeError):
                pass
        for proc in self._workers:
            proc.join(timeout=1.0)
        self._links, self._workers = [], []

    def __del__(self):
        self.close()
