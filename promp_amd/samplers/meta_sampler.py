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
import operator
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

    def cut(self, env, slabs=None, task=0):
        """the finished episode of environment `env` as a path dict; its slot starts over.  With `slabs` (_EpisodeSlabs) the
        fields that go to the device leave the tables ONCE, straight into their final position of the meta-batch's flat arrays,
        and the path dict holds views of those rows"""
        n = int(self.fill[env])
        path = {'env_infos': {}, 'agent_infos': {}}
        dest = slabs.place(task, n, self._tables) if slabs is not None else None
        for key_path, table in self._tables.items():
            if dest is not None and key_path in dest:
                view = dest[key_path]
                view[...] = table[env, :n]
                _graft(path, key_path, view)
            else:
                _graft(path, key_path, table[env, :n].copy())
        self.fill[env] = 0
        if dest is not None:
            slabs.remember(task, path, n)
        return path, n


_PATH_FIELDS = operator.itemgetter('observations', 'actions', 'rewards', 'agent_infos')
_INFO_FIELDS = operator.itemgetter('mean', 'log_std')


class HostPaths(OrderedDict):
    """MetaSampler.obtain_samples' return value (OrderedDict{task -> [path dicts]}, meta_sampler.py:59-137) whose device-bound
    fields -- observations, actions, rewards, agent_infos mean / log_std -- are VIEWS of flat [rows, dim] arrays in task-major,
    path-major order: `flat` is what promp_amd._lib.flatten_paths would build from the dicts, already built (in page-locked
    memory when the library is there).  process_samples then uploads it as it is instead of concatenating 800 paths x 5 arrays
    again.  The dicts stay ordinary dicts; flat_if_intact() hands `flat` out only while every path still holds the very arrays
    the sampler put there (a caller that replaces a path's rewards, drops a path, ... gets the general route)."""
    flat = None
    _origin = ()

    def __reduce__(self):
        # A copy (copy.copy / copy.deepcopy / pickle: replay buffers, worker processes) is a HostPaths WITHOUT `flat`: the arrays
        # of its path dicts no longer alias the flat arrays, so an in-place edit of the copy would otherwise be ignored in favour
        # of the stale flat arrays (the identities flat_if_intact compares survive a pickle round trip).  The copy takes the
        # general route; slab_backed() re-points it if the fast one is wanted.
        return (HostPaths, (), None, None, iter(OrderedDict.items(self)))

    def flat_if_intact(self):
        fl, org = self.flat, self._origin
        if fl is None:
            return None
        # (one C-level lookup of the four fields per path, one of the two agent_infos entries: at 800 paths the .get calls of the
        #  obvious loop were a third of a millisecond per sampling step)
        fields, infos = _PATH_FIELDS, _INFO_FIELDS
        i, n = 0, len(org)
        try:
            for plist in OrderedDict.values(self):
                for p in plist:
                    o = org[i]
                    i += 1
                    obs, act, rew, ai = fields(p)
                    mean, log_std = infos(ai)
                    if o[0] is not p or obs is not o[1] or act is not o[2] or rew is not o[3] or mean is not o[4] or log_std is not o[5]:
                        return None
        except (IndexError, KeyError, TypeError):       # more paths than the sampler placed, a field dropped, agent_infos replaced
            return None
        return fl if i == n else None


class _EpisodeSlabs(object):
    """The flat arrays of a meta-batch under construction: task i owns rows [i cap, (i + 1) cap) of every array, cap =
    rollouts_per_meta_task x max_path_length (what a task contributes when no episode ends early: then the arrays are final
    the moment the last episode is cut, and nothing is ever copied again).  Environments that end episodes early fill the tasks
    unevenly (collection stops on the TOTAL of finished steps): a task that outgrows its share, or tables that are not float32
    (an environment with float64 observations keeps its dtype in the path dicts, as the reference does), switch the slabs off
    for this batch -- the paths are then what they always were, independent arrays, and process_samples flattens them."""
    FIELDS = (('observations',), ('actions',), ('rewards',), ('agent_infos', 'mean'), ('agent_infos', 'log_std'))
    NAMES = ('obs', 'act', 'rew', 'old_mean', 'old_log_std')

    def __init__(self, n_tasks, cap):
        self.M, self.cap = int(n_tasks), int(cap)
        self.cursor = [0] * self.M
        self.bufs = None
        self.ok = True
        self.entries = [[] for _ in range(self.M)]        # per task: (path dict, n)

    def _allocate(self, tables):
        from .. import _lib
        try:
            lib = _lib.get_library()
        except Exception:
            lib = None
        bufs = {}
        for key in self.FIELDS:
            t = tables.get(key)
            if t is None or (t.dtype != np.float32 and not (key == ('rewards',) and t.dtype == np.float64)):
                return None
            if key == ('rewards',):
                if t.ndim != 2:
                    return None
                bufs[key] = _lib.host_pool.get((self.M * self.cap,), t.dtype, lib)
            else:
                if t.ndim != 3:
                    return None
                bufs[key] = _lib.host_pool.get((self.M * self.cap, t.shape[2]), np.float32, lib)
        return bufs

    def place(self, task, n, tables):
        if not self.ok:
            return None
        if self.bufs is None:
            self.bufs = self._allocate(tables)
            if self.bufs is None:
                self.ok = False
                return None
        if self.cursor[task] + n > self.cap or any(tables[k].dtype != self.bufs[k].dtype for k in self.FIELDS):
            self.ok = False                       # (paths cut so far keep their views: valid arrays, just no shared flat any more)
            return None
        a = task * self.cap + self.cursor[task]
        self.cursor[task] += n
        return {k: self.bufs[k][a:a + n] for k in self.FIELDS}

    def remember(self, task, path, n):
        self.entries[task].append((path, n))

    def finish(self, paths):
        """-> HostPaths with .flat, or a plain OrderedDict when the slabs were switched off"""
        if not self.ok or self.bufs is None or any(len(paths[i]) != len(self.entries[i]) for i in range(self.M)):
            return paths
        rows = sum(self.cursor)
        if any(c != self.cap for c in self.cursor):
            # Episodes ended early.  Collection stops at M x cap finished steps in all, so a task below its share means another
            # one above it (which switched the slabs off in place()) -- or an interrupted collection: the general route either way
            # (the views the paths hold are valid arrays on their own).
            return paths
        out = HostPaths(paths)
        lens = [n for i in range(self.M) for _, n in self.entries[i]]
        pro = np.zeros(len(lens) + 1, np.int32)
        np.cumsum(lens, out=pro[1:])
        tpo = np.zeros(self.M + 1, np.int32)
        np.cumsum([len(self.entries[i]) for i in range(self.M)], out=tpo[1:])
        flat = dict(task_path_offsets=tpo, path_row_offsets=pro)
        for k, name in zip(self.FIELDS, self.NAMES):
            flat[name] = self.bufs[k][:rows]
        out.flat = flat
        out._origin = [(p, p['observations'], p['actions'], p['rewards'], p['agent_infos']['mean'], p['agent_infos']['log_std'])
                       for i in range(self.M) for p, _ in self.entries[i]]
        return out


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
        slabs = _EpisodeSlabs(M, self.batch_size * self.max_path_length)
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
                path, n = tables.cut(env, slabs, env // per_task)
                paths[env // per_task].append(path)
                collected += n
            observations = next_observations
        self.total_timesteps_sampled += self.total_samples
        if log:
            logger.logkv(log_prefix + 'PolicyExecTime', policy_seconds)
            logger.logkv(log_prefix + 'EnvExecTime', env_seconds)
        return slabs.finish(paths)


    def _per_env(self, infos_by_task):
        """agent infos arrive as list[task] of list[env of the task]; flatten to environment order"""
        if not infos_by_task:
            return []
        assert len(infos_by_task) == self.meta_batch_size and all(len(t) == self.envs_per_task for t in infos_by_task)
        return [info for task_infos in infos_by_task for info in task_infos]


def slab_backed(paths_meta_batch):
    """Ordinary path dicts -> the HostPaths MetaSampler.obtain_samples builds (one copy of the device-bound fields into flat
    arrays, the dicts re-pointed at views of them); returns the input unchanged when the fields are not float32 arrays.
    For callers that assemble path dicts themselves (a custom sampler, replayed trajectories, bench.py's synthetic batch)."""
    from .. import _lib
    M = len(paths_meta_batch)
    plists = list(paths_meta_batch.values())
    try:
        fl = _lib.flatten_paths(paths_meta_batch, _lib.get_library())
    except Exception:
        return paths_meta_batch
    first = plists[0][0]
    if fl['act'] is None or any(np.asarray(x).dtype != np.float32 for x in (first['observations'], first['actions'], first['agent_infos']['mean'],
                                                                             first['agent_infos']['log_std'])):
        return paths_meta_batch
    out = HostPaths((k, v) for k, v in paths_meta_batch.items())
    pro = fl['path_row_offsets']
    i = 0
    origin = []
    for plist in plists:
        for p in plist:
            a, b = int(pro[i]), int(pro[i + 1])
            i += 1
            p['observations'], p['actions'], p['rewards'] = fl['obs'][a:b], fl['act'][a:b], fl['rew'][a:b]
            p['agent_infos']['mean'], p['agent_infos']['log_std'] = fl['old_mean'][a:b], fl['old_log_std'][a:b]
            origin.append((p, p['observations'], p['actions'], p['rewards'], p['agent_infos']['mean'], p['agent_infos']['log_std']))
    out.flat, out._origin = fl, origin
    return out
