"""MetaSampler (reference: meta_policy_search/samplers/meta_sampler.py:11-155): rollout collection, host side."""
import itertools
import time
from collections import OrderedDict

import numpy as np

from ..utils import logger
from .vectorized_env_executor import MetaIterativeEnvExecutor, MetaParallelEnvExecutor


def _stack_tensor_dict_list(dict_list):
    """utils/utils.py:125-143"""
    if not dict_list:
        return {}
    out = {}
    for k in dict_list[0].keys():
        ex = dict_list[0][k]
        out[k] = _stack_tensor_dict_list([d[k] for d in dict_list]) if isinstance(ex, dict) \
            else np.asarray([d[k] for d in dict_list])
    return out


class MetaSampler(object):
    """
    Args (meta_sampler.py:25-49): env, policy, rollouts_per_meta_task, meta_batch_size, max_path_length,
    envs_per_task=None, parallel=False
    """

    def __init__(self, env, policy, rollouts_per_meta_task, meta_batch_size, max_path_length, envs_per_task=None,
                 parallel=False):
        assert hasattr(env, 'reset') and hasattr(env, 'step')
        assert hasattr(env, 'set_task')
        self.env, self.policy = env, policy
        self.batch_size = rollouts_per_meta_task
        self.max_path_length = max_path_length
        self.envs_per_task = rollouts_per_meta_task if envs_per_task is None else envs_per_task
        self.meta_batch_size = meta_batch_size
        self.total_samples = meta_batch_size * rollouts_per_meta_task * max_path_length
        self.parallel = parallel
        self.total_timesteps_sampled = 0
        cls = MetaParallelEnvExecutor if parallel else MetaIterativeEnvExecutor
        self.vec_env = cls(env, self.meta_batch_size, self.envs_per_task, self.max_path_length)

    def update_tasks(self):
        tasks = self.env.sample_tasks(self.meta_batch_size)
        assert len(tasks) == self.meta_batch_size
        self.vec_env.set_tasks(tasks)

    def obtain_samples(self, log=False, log_prefix=''):
        """-> OrderedDict{task -> list of path dicts}  (meta_sampler.py:59-137)"""
        paths = OrderedDict((i, []) for i in range(self.meta_batch_size))
        n_samples = 0
        running_paths = [_get_empty_running_paths_dict() for _ in range(self.vec_env.num_envs)]
        policy_time, env_time = 0, 0
        policy = self.policy
        obses = self.vec_env.reset()
        while n_samples < self.total_samples:
            t = time.time()
            obs_per_task = np.split(np.asarray(obses), self.meta_batch_size)
            actions, agent_infos = policy.get_actions(obs_per_task)
            policy_time += time.time() - t
            t = time.time()
            actions = np.concatenate(actions)
            next_obses, rewards, dones, env_infos = self.vec_env.step(actions)
            env_time += time.time() - t
            agent_infos, env_infos = self._handle_info_dicts(agent_infos, env_infos)
            new_samples = 0
            for idx, observation, action, reward, env_info, agent_info, done in zip(
                    itertools.count(), obses, actions, rewards, env_infos, agent_infos, dones):
                rp = running_paths[idx]
                rp['observations'].append(observation)
                rp['actions'].append(action)
                rp['rewards'].append(reward)
                rp['env_infos'].append(env_info)
                rp['agent_infos'].append(agent_info)
                if done:
                    paths[idx // self.envs_per_task].append(dict(
                        observations=np.asarray(rp['observations']), actions=np.asarray(rp['actions']),
                        rewards=np.asarray(rp['rewards']), env_infos=_stack_tensor_dict_list(rp['env_infos']),
                        agent_infos=_stack_tensor_dict_list(rp['agent_infos'])))
                    new_samples += len(rp['rewards'])
                    running_paths[idx] = _get_empty_running_paths_dict()
            n_samples += new_samples
            obses = next_obses
        self.total_timesteps_sampled += self.total_samples
        if log:
            logger.logkv(log_prefix + 'PolicyExecTime', policy_time)
            logger.logkv(log_prefix + 'EnvExecTime', env_time)
        return paths

    def _handle_info_dicts(self, agent_infos, env_infos):
        if not env_infos:
            env_infos = [dict() for _ in range(self.vec_env.num_envs)]
        if not agent_infos:
            agent_infos = [dict() for _ in range(self.vec_env.num_envs)]
        else:
            assert len(agent_infos) == self.meta_batch_size
            assert len(agent_infos[0]) == self.envs_per_task
            agent_infos = sum(agent_infos, [])
        assert len(agent_infos) == self.meta_batch_size * self.envs_per_task == len(env_infos)
        return agent_infos, env_infos


def _get_empty_running_paths_dict():
    return dict(observations=[], actions=[], rewards=[], env_infos=[], agent_infos=[])
