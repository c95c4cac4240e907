"""Rollout collection for a meta-batch of tasks.

Same constructor, `update_tasks()` and `obtain_samples()` contract as the reference's MetaSampler
(meta_policy_search/samplers/meta_sampler.py:25-137):

    sampler = MetaSampler(env, policy, rollouts_per_meta_task, meta_batch_size, max_path_length, envs_per_task=None, parallel=False)
    sampler.update_tasks()
    paths = sampler.obtain_samples(log=False, log_prefix='')
        -> OrderedDict{task index -> [path, ...]},  path = dict(observations [T,O], actions [T,A], rewards [T],
                                                                  env_infos{...}, agent_infos{mean [T,A], log_std [T,A]})

Collection stops once meta_batch_size * rollouts_per_meta_task * max_path_length environment steps belong to FINISHED
trajectories; trajectories still running at that point are dropped.

Design.  The environments run in lock step, so a rollout is a set of dense [environment, time] tables.  `_StepTables`
preallocates them once per sampler (the pool never lets an episode exceed max_path_length) and every environment step is
ONE vectorised write of a column into each table; a finished episode is cut out as a slice.  This is also the layout of
the device slabs ([task x path x t] rows), which is what lets a device-side writer fill the same positions without a
host copy (samplers/device_point_sampler.py does so for the point environment).
"""
import time
from collections import OrderedDict

import numpy as np

from ..utils import logger
from .vectorized_env_executor import MetaIterativeEnvExecutor, MetaParallelEnvExecutor


def _leaves(tree, prefix=()):
    """(key path, value) for every leaf of a (possibly nested) info dict"""
    for key, value in tree.items():
        if isinstance(value, dict):
            yield from _leaves(value, prefix + (key,))
        else:
            yield prefix + (key,), value


def _graft(tree, key_path, value):
    for key in key_path[:-1]:
        tree = tree.setdefault(key, {})
    tree[key_path[-1]] = value


class _StepTables(object):
    """[environment, time, ...] tables of one rollout, written a column per environment step.

    A table is allocated the first time its key shows up (its per-step shape is taken from that first value).  Its dtype
    follows NumPy's promotion over everything written so far, as `np.asarray` over the reference's per-step lists does
    (meta_sampler.py:100-125 appends Python scalars and converts at the end): an environment that returns the int 0 on some
    steps and floats on others (MetaPointEnvCorner's sparse reward) gets a float table, not a truncating int one.
    `fill[e]` is the length of environment e's episode in progress."""

    def __init__(self, n_envs, horizon):
        self.n_envs, self.horizon = int(n_envs), int(horizon)
        self.fill = np.zeros(self.n_envs, dtype=np.int64)
        self._tables = {}                       # key path (tuple) -> ndarray [n_envs, horizon, ...]
        self._rows = np.arange(self.n_envs)

    def _table(self, key_path, column):
        """the table of `key_path`, wide enough in dtype for `column` ([n_envs, ...])"""
        table = self._tables.get(key_path)
        if table is None:
            table = np.zeros((self.n_envs, self.horizon) + column.shape[1:], dtype=column.dtype)
            self._tables[key_path] = table
        elif table.dtype != column.dtype:
            wide = np.result_type(table.dtype, column.dtype)
            if wide != table.dtype:
                table = table.astype(wide)
                self._tables[key_path] = table
        return table

    def put(self, key_path, column):
        """column: one value per environment, for the current time step of each"""
        column = np.asarray(column)
        self._table(key_path, column)[self._rows, self.fill] = column

    def put_infos(self, group, infos):
        """infos: one (possibly nested, possibly empty) dict per environment"""
        if not infos or not infos[0]:
            return
        for key_path, _ in _leaves(infos[0]):
            values = []
            for info in infos:
                node = info
                for key in key_path:
                    node = node[key]
                values.append(node)
            column = np.asarray(values)
            self._table((group,) + key_path, column)[self._rows, self.fill] = column

    def advance(self):
        self.fill += 1

    def cut(self, env):
        """the finished episode of environment `env` as a path dict; its slot starts over"""
        n = int(self.fill[env])
        path = {'env_infos': {}, 'agent_infos': {}}
        for key_path, table in self._tables.items():
            _graft(path, key_path, table[env, :n].copy())
        self.fill[env] = 0
        return path, n


class MetaSampler(object):
    """
    Args: env (needs reset / step / set_task / sample_tasks), policy (get_actions), rollouts_per_meta_task,
    meta_batch_size, max_path_length, envs_per_task=None (defaults to rollouts_per_meta_task),
    parallel=False (one worker process per task)
    """

    def __init__(self, env, policy, rollouts_per_meta_task, meta_batch_size, max_path_length, envs_per_task=None,
                 parallel=False):
        missing = [name for name in ('reset', 'step', 'set_task') if not hasattr(env, name)]
        assert not missing, 'environment lacks %s' % ', '.join(missing)
        self.env, self.policy = env, policy
        self.batch_size = rollouts_per_meta_task
        self.meta_batch_size = meta_batch_size
        self.max_path_length = max_path_length
        self.envs_per_task = rollouts_per_meta_task if envs_per_task is None else envs_per_task
        self.total_samples = meta_batch_size * rollouts_per_meta_task * max_path_length
        self.total_timesteps_sampled = 0
        self.parallel = parallel
        pool_class = MetaParallelEnvExecutor if parallel else MetaIterativeEnvExecutor
        self.vec_env = pool_class(env, meta_batch_size, self.envs_per_task, max_path_length)

    def update_tasks(self):
        """one freshly drawn task per meta-batch slot"""
        tasks = self.env.sample_tasks(self.meta_batch_size)
        assert len(tasks) == self.meta_batch_size
        self.vec_env.set_tasks(tasks)

    def obtain_samples(self, log=False, log_prefix=''):
        M, per_task = self.meta_batch_size, self.envs_per_task
        tables = _StepTables(self.vec_env.num_envs, self.max_path_length)
        paths = OrderedDict((task, []) for task in range(M))
        collected, policy_seconds, env_seconds = 0, 0.0, 0.0
        observations = self.vec_env.reset()
        while collected < self.total_samples:
            started = time.time()
            obs_by_task = np.asarray(observations).reshape((M, per_task) + np.shape(observations[0]))
            actions_by_task, infos_by_task = self.policy.get_actions(list(obs_by_task))
            actions = np.concatenate(actions_by_task)
            policy_seconds += time.time() - started

            started = time.time()
            next_observations, rewards, finished, env_infos = self.vec_env.step(actions)
            env_seconds += time.time() - started

            tables.put(('observations',), observations)
            tables.put(('actions',), actions)
            tables.put(('rewards',), rewards)
            tables.put_infos('env_infos', env_infos)
            tables.put_infos('agent_infos', self._per_env(infos_by_task))
            tables.advance()
            for env in np.flatnonzero(np.asarray(finished)):
                path, n = tables.cut(env)
                paths[env // per_task].append(path)
                collected += n
            observations = next_observations
        self.total_timesteps_sampled += self.total_samples
        if log:
            logger.logkv(log_prefix + 'PolicyExecTime', policy_seconds)
            logger.logkv(log_prefix + 'EnvExecTime', env_seconds)
        return paths

    def _per_env(self, infos_by_task):
        """agent infos arrive as list[task] of list[env of the task]; flatten to environment order"""
        if not infos_by_task:
            return []
        assert len(infos_by_task) == self.meta_batch_size and all(len(t) == self.envs_per_task for t in infos_by_task)
        return [info for task_infos in infos_by_task for info in task_infos]
