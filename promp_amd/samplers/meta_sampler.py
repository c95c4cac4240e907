"""Rollout collection for a meta-batch of tasks, host side.

Same constructor, `update_tasks()` and `obtain_samples()` contract as the reference's MetaSampler
(meta_policy_search/samplers/meta_sampler.py:25-137): `envs_per_task` environments per task are stepped in lock step,
every finished trajectory is filed under its task, and collection stops once
meta_batch_size * rollouts_per_meta_task * max_path_length environment steps have been filed (trajectories still running
at that point are dropped).  The policy is queried once per environment step for all tasks at once
(`policy.get_actions`, the mean network runs on the device).
"""
import time
from collections import OrderedDict

import numpy as np

from ..utils import logger
from .vectorized_env_executor import MetaIterativeEnvExecutor, MetaParallelEnvExecutor


def _stack_infos(records):
    """list of (possibly nested) dicts -> dict of stacked arrays (what utils.stack_tensor_dict_list yields)"""
    if len(records) == 0:
        return {}
    stacked = {}
    for key, first in records[0].items():
        column = [rec[key] for rec in records]
        stacked[key] = _stack_infos(column) if isinstance(first, dict) else np.asarray(column)
    return stacked


class _Trajectory(object):
    """one environment's trajectory in progress"""
    __slots__ = ('obs', 'act', 'rew', 'env_info', 'agent_info')

    def __init__(self):
        self.obs, self.act, self.rew, self.env_info, self.agent_info = [], [], [], [], []

    def record(self, obs, act, rew, env_info, agent_info):
        self.obs.append(obs)
        self.act.append(act)
        self.rew.append(rew)
        self.env_info.append(env_info)
        self.agent_info.append(agent_info)

    def __len__(self):
        return len(self.rew)

    def as_path(self):
        return dict(observations=np.asarray(self.obs), actions=np.asarray(self.act), rewards=np.asarray(self.rew),
                    env_infos=_stack_infos(self.env_info), agent_infos=_stack_infos(self.agent_info))


class MetaSampler(object):
    """
    Args: env (needs reset / step / set_task / sample_tasks), policy, rollouts_per_meta_task, meta_batch_size,
    max_path_length, envs_per_task=None (defaults to rollouts_per_meta_task), parallel=False (worker processes)
    """

    def __init__(self, env, policy, rollouts_per_meta_task, meta_batch_size, max_path_length, envs_per_task=None,
                 parallel=False):
        for needed in ('reset', 'step', 'set_task'):
            assert hasattr(env, needed)
        self.env = env
        self.policy = policy
        self.batch_size = rollouts_per_meta_task
        self.meta_batch_size = meta_batch_size
        self.max_path_length = max_path_length
        self.envs_per_task = envs_per_task if envs_per_task is not None else rollouts_per_meta_task
        self.total_samples = meta_batch_size * rollouts_per_meta_task * max_path_length
        self.total_timesteps_sampled = 0
        self.parallel = parallel
        executor = MetaParallelEnvExecutor if parallel else MetaIterativeEnvExecutor
        self.vec_env = executor(env, meta_batch_size, self.envs_per_task, max_path_length)

    def update_tasks(self):
        """draw one task per meta-batch slot and hand them to the environments"""
        tasks = self.env.sample_tasks(self.meta_batch_size)
        assert len(tasks) == self.meta_batch_size
        self.vec_env.set_tasks(tasks)

    def obtain_samples(self, log=False, log_prefix=''):
        """-> OrderedDict{task index -> [path, ...]}, path = dict(observations [T,O], actions [T,A], rewards [T],
        env_infos, agent_infos{mean [T,A], log_std [T,A]})"""
        n_envs, per_task = self.vec_env.num_envs, self.envs_per_task
        filed = OrderedDict((task, []) for task in range(self.meta_batch_size))
        in_progress = [_Trajectory() for _ in range(n_envs)]
        steps_filed, seconds = 0, dict(policy=0.0, env=0.0)
        observations = self.vec_env.reset()
        while steps_filed < self.total_samples:
            tick = time.time()
            actions, agent_infos = self.policy.get_actions(np.split(np.asarray(observations), self.meta_batch_size))
            actions = np.concatenate(actions)
            seconds['policy'] += time.time() - tick
            tick = time.time()
            next_observations, rewards, dones, env_infos = self.vec_env.step(actions)
            seconds['env'] += time.time() - tick
            agent_infos, env_infos = self._handle_info_dicts(agent_infos, env_infos)
            for e in range(n_envs):
                traj = in_progress[e]
                traj.record(observations[e], actions[e], rewards[e], env_infos[e], agent_infos[e])
                if dones[e]:
                    filed[e // per_task].append(traj.as_path())
                    steps_filed += len(traj)
                    in_progress[e] = _Trajectory()
            observations = next_observations
        self.total_timesteps_sampled += self.total_samples
        if log:
            logger.logkv(log_prefix + 'PolicyExecTime', seconds['policy'])
            logger.logkv(log_prefix + 'EnvExecTime', seconds['env'])
        return filed

    def _handle_info_dicts(self, agent_infos, env_infos):
        """one info dict per environment, in environment order (agent infos arrive grouped by task)"""
        n_envs = self.vec_env.num_envs
        env_infos = list(env_infos) if env_infos else [dict() for _ in range(n_envs)]
        if agent_infos:
            assert len(agent_infos) == self.meta_batch_size and len(agent_infos[0]) == self.envs_per_task
            agent_infos = [info for task_infos in agent_infos for info in task_infos]
        else:
            agent_infos = [dict() for _ in range(n_envs)]
        assert len(agent_infos) == len(env_infos) == n_envs
        return agent_infos, env_infos
