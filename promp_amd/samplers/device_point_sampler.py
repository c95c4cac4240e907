"""Rollouts of the 2-D point-mass meta-environment collected ON THE DEVICE.

Same interface as MetaSampler (update_tasks / obtain_samples / total_timesteps_sampled), for the environment of BASELINE
config 1: promp_amd.envs.point_env.MetaPointEnvCorner, bare or wrapped in promp_amd.envs.normalized_env.normalize.  One kernel launch runs every environment for the whole horizon under its task's
current policy parameters and writes the trajectories straight into the sampling step's slab
(promp_rollout_point_env); MetaSampleProcessor.process_samples then finds the data resident and uploads nothing.
The host keeps what it is good at: drawing tasks, start states and the exploration noise from NumPy's RNG, and
materialising the path dicts the plugin API returns (one small download per sampling step)."""
from collections import OrderedDict

import numpy as np

from ..utils import logger


class DevicePaths(OrderedDict):
    """{task -> [path dicts]} as obtain_samples returns, plus where the same data already lives on the device.

    process_samples leaves its per-path side effect (every path dict gains 'returns' and 'advantages', samplers/base.py:104,159
    of the reference) PENDING on a container of this kind: it is applied the first time anybody looks at the path lists again
    (1 600 dict stores per sampling step that the training loop never reads)."""
    device_ref = None     # (session serial, upload serial, step slot)
    flat = None           # dict(task_path_offsets, path_row_offsets) of the resident slab
    _pending = None       # callable applying the side effect, or None

    def settle(self):
        todo, self._pending = self._pending, None
        if todo is not None:
            todo()

    def raw_values(self):
        """the path lists without settling (library-internal)"""
        return OrderedDict.values(self)

    def __getitem__(self, key):
        self.settle()
        return OrderedDict.__getitem__(self, key)

    def get(self, key, default=None):
        self.settle()
        return OrderedDict.get(self, key, default)

    def values(self):
        self.settle()
        return OrderedDict.values(self)

    def items(self):
        self.settle()
        return OrderedDict.items(self)

    def pop(self, *args):
        self.settle()
        return OrderedDict.pop(self, *args)

    def popitem(self, *args, **kw):
        self.settle()
        return OrderedDict.popitem(self, *args, **kw)


def _point_env_options(env):
    """(bare environment, kernel options) of MetaPointEnvCorner, wrapped in normalize(...) or not"""
    from ..envs.normalized_env import NormalizedEnv
    from ..envs.point_env import MetaPointEnvCorner
    scale = 0.0                              # bare environment
    if isinstance(env, NormalizedEnv):
        assert not (env._normalize_obs or env._normalize_reward), 'running observation / reward normalisation is host-side state'
        bare = env.wrapped_env
        scale = float(env._normalization_scale)
        assert np.allclose(bare.action_space.low, -bare.action_space.high)
    else:
        bare = env
    assert isinstance(bare, MetaPointEnvCorner), 'the device environment is MetaPointEnvCorner (optionally normalize()d)'
    return bare, dict(reward_type=bare.reward_type, normalization_scale=scale, max_step=float(bare.action_space.high[0]),
                      sparse_radius=float(bare.sparse_reward_radius))


class DevicePointEnvSampler(object):
    """device_noise=False: start states and exploration noise come from NumPy's RNG (reproducible from np.random.seed, and
    comparable step by step with the NumPy environment); True: the noise is drawn on the device (Philox4x32-10 keyed by a
    seed taken from NumPy's RNG once per sampling step), nothing but goals and start states is uploaded."""

    def __init__(self, env, policy, rollouts_per_meta_task, meta_batch_size, max_path_length, envs_per_task=None,
                 parallel=False, device_noise=False):
        assert hasattr(env, 'sample_tasks') and hasattr(env, 'set_task')
        assert policy.obs_dim == 2 and policy.action_dim == 2, 'the device environment is the 2-D point mass'
        self.env, self.policy = env, policy
        self._bare, self._env_opts = _point_env_options(env)
        self.device_noise = device_noise
        self.batch_size = rollouts_per_meta_task
        self.meta_batch_size = meta_batch_size
        self.max_path_length = max_path_length
        self.envs_per_task = rollouts_per_meta_task       # one environment per rollout: every path runs the full horizon
        self.total_samples = meta_batch_size * rollouts_per_meta_task * max_path_length
        self.total_timesteps_sampled = 0
        self.goals = None

    def update_tasks(self):
        tasks = self.env.sample_tasks(self.meta_batch_size)
        assert len(tasks) == self.meta_batch_size
        self.goals = np.asarray(tasks, dtype=np.float64).reshape(self.meta_batch_size, 2)

    def obtain_samples(self, log=False, log_prefix=''):
        assert self.goals is not None, 'call update_tasks() first'
        M, B, T = self.meta_batch_size, self.envs_per_task, self.max_path_length
        sess = self.policy.session
        ctx = sess.ensure(M * B * T, M * B)
        if sess.task_thetas is not None:           # parameters set while no context existed yet
            ctx.set_task_thetas(sess.task_thetas)
            sess.task_thetas = None
        start = np.random.uniform(-0.2, 0.2, size=(M, B, 2))         # MetaPointEnvCorner.reset
        slot = sess.next_slot()
        if self.device_noise:
            ctx.rollout_point_env(slot, self.goals, start, None, clip_infos=self.policy._pre_update_mode, path_length=T,
                                  seed=int(np.random.randint(0, 2 ** 31 - 1)), **self._env_opts)
        else:
            noise = np.random.normal(size=(M, B, T, 2)).astype(np.float32)
            ctx.rollout_point_env(slot, self.goals, start, noise, clip_infos=self.policy._pre_update_mode, **self._env_opts)
        sess._upload_counter += 1
        sess.upload_serial[slot] = sess._upload_counter
        slab = ctx.download_step(slot)
        paths = DevicePaths()
        for i in range(M):
            paths[i] = []
            log_std = np.tile(slab['old_log_std'][i], (T, 1))
            for b in range(B):
                rows = slice((i * B + b) * T, (i * B + b + 1) * T)
                paths[i].append(dict(observations=slab['obs'][rows], actions=slab['act'][rows], rewards=slab['rew'][rows],
                                     env_infos={},
                                     agent_infos=dict(mean=slab['old_mean'][rows], log_std=log_std)))
        paths.device_ref = (sess.serial, sess.upload_serial[slot], slot)
        paths.flat = dict(task_path_offsets=np.arange(M + 1, dtype=np.int32) * B,
                          path_row_offsets=np.arange(M * B + 1, dtype=np.int32) * T,
                          obs=slab['obs'], act=slab['act'], rew=slab['rew'], old_mean=slab['old_mean'], old_log_std=slab['old_log_std'])
        self.total_timesteps_sampled += M * B * T
        if log:
            logger.logkv(log_prefix + 'PolicyExecTime', 0.0)
            logger.logkv(log_prefix + 'EnvExecTime', 0.0)
        return paths
