"""Vectorised environment executors (reference: meta_policy_search/samplers/vectorized_env_executor.py:7-234).
Host-side, as in the reference: environment physics is third-party Python/C and not part of the hot path."""
import copy
import pickle as pickle
from multiprocessing import Pipe, Process

import numpy as np


class MetaIterativeEnvExecutor(object):
    """meta_batch_size * envs_per_task deep copies of the env stepped in a loop (vectorized_env_executor.py:7-85)"""

    def __init__(self, env, meta_batch_size, envs_per_task, max_path_length):
        self.envs = np.asarray([copy.deepcopy(env) for _ in range(meta_batch_size * envs_per_task)])
        self.ts = np.zeros(len(self.envs), dtype='int')
        self.max_path_length = max_path_length

    def step(self, actions):
        assert len(actions) == self.num_envs
        all_results = [env.step(a) for (a, env) in zip(actions, self.envs)]
        obs, rewards, dones, env_infos = list(map(list, zip(*all_results)))
        dones = np.asarray(dones)
        self.ts += 1
        dones = np.logical_or(self.ts >= self.max_path_length, dones)
        for i in np.argwhere(dones).flatten():
            obs[i] = self.envs[i].reset()
            self.ts[i] = 0
        return obs, rewards, dones, env_infos

    def set_tasks(self, tasks):
        envs_per_task = np.split(self.envs, len(tasks))
        for task, envs in zip(tasks, envs_per_task):
            for env in envs:
                env.set_task(task)

    def reset(self):
        obses = [env.reset() for env in self.envs]
        self.ts[:] = 0
        return obses

    @property
    def num_envs(self):
        return len(self.envs)


def worker(remote, parent_remote, env_pickle, n_envs, max_path_length, seed):
    """one worker process per meta-task (vectorized_env_executor.py:180-234)"""
    parent_remote.close()
    envs = [pickle.loads(env_pickle) for _ in range(n_envs)]
    np.random.seed(seed)
    ts = np.zeros(n_envs, dtype='int')
    while True:
        cmd, data = remote.recv()
        if cmd == 'step':
            all_results = [env.step(a) for (a, env) in zip(data, envs)]
            obs, rewards, dones, infos = map(list, zip(*all_results))
            ts += 1
            for i in range(n_envs):
                if dones[i] or (ts[i] >= max_path_length):
                    dones[i] = True
                    obs[i] = envs[i].reset()
                    ts[i] = 0
            remote.send((obs, rewards, dones, infos))
        elif cmd == 'reset':
            obs = [env.reset() for env in envs]
            ts[:] = 0
            remote.send(obs)
        elif cmd == 'set_task':
            for env in envs:
                env.set_task(data)
            remote.send(None)
        elif cmd == 'close':
            remote.close()
            break
        else:
            raise NotImplementedError


class MetaParallelEnvExecutor(object):
    """one daemon worker process per meta-task, pipes (vectorized_env_executor.py:88-177)"""

    def __init__(self, env, meta_batch_size, envs_per_task, max_path_length):
        self.n_envs = meta_batch_size * envs_per_task
        self.meta_batch_size = meta_batch_size
        self.envs_per_task = envs_per_task
        self.remotes, self.work_remotes = zip(*[Pipe() for _ in range(meta_batch_size)])
        seeds = np.random.choice(range(10 ** 6), size=meta_batch_size, replace=False)
        self.ps = [Process(target=worker, args=(wr, r, pickle.dumps(env), envs_per_task, max_path_length, seed))
                   for (wr, r, seed) in zip(self.work_remotes, self.remotes, seeds)]
        for p in self.ps:
            p.daemon = True
            p.start()
        for remote in self.work_remotes:
            remote.close()

    def step(self, actions):
        assert len(actions) == self.num_envs
        chunks = [actions[i:i + self.envs_per_task] for i in range(0, len(actions), self.envs_per_task)]
        for remote, a in zip(self.remotes, chunks):
            remote.send(('step', a))
        results = [remote.recv() for remote in self.remotes]
        obs, rewards, dones, env_infos = map(lambda x: sum(x, []), zip(*results))
        return obs, rewards, dones, env_infos

    def reset(self):
        for remote in self.remotes:
            remote.send(('reset', None))
        return sum([remote.recv() for remote in self.remotes], [])

    def set_tasks(self, tasks=None):
        for remote, task in zip(self.remotes, tasks):
            remote.send(('set_task', task))
        for remote in self.remotes:
            remote.recv()

    @property
    def num_envs(self):
        return self.n_envs
