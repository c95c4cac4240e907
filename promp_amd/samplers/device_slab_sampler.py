"""MetaSampler for host-stepped environments (MuJoCo ...) whose policy queries fill the sampling step's slab ON THE DEVICE
(SURVEY.md 8f row 1).

The reference pays, per environment step, one sess.run with the observations fed and the means fetched, host-side noise,
and at the end of the rollout a re-upload of everything as feed dicts (policies/meta_gaussian_mlp_policy.py:99-157,
samplers/meta_sampler.py:87-125, meta_algos/base.py:245-301).  Here a step is one small call:

    actions = ctx.policy_step(slot, t, observations)      # [M, B, O] in, [M, B, A] out

the device evaluates every task's mean network, draws the exploration noise (Philox4x32-10, keyed by a seed taken from
NumPy's RNG once per sampling step) and writes observation, action and mean into the slab at row (task, env, t).  After the
last step the rewards follow in one upload (promp_set_rewards) and MetaSampleProcessor.process_samples finds the data
resident: no trajectory upload at all.  The path dicts of the plugin API are materialised from one download per sampling
step.

The slab layout is one fixed-length path per environment, so this sampler needs envs_per_task == rollouts_per_meta_task
and environments that run to the horizon; if an episode ends early the sampling step is collected again by the host-side
MetaSampler logic (which handles ragged paths), so the result is always a valid set of paths.
"""
import time

import numpy as np

from ..utils import logger
from .device_point_sampler import DevicePaths
from .meta_sampler import MetaSampler, _StepTables


class DeviceSlabSampler(MetaSampler):
    def __init__(self, env, policy, rollouts_per_meta_task, meta_batch_size, max_path_length, envs_per_task=None,
                 parallel=False):
        assert envs_per_task in (None, rollouts_per_meta_task), 'one environment per rollout (fixed-length slab rows)'
        super(DeviceSlabSampler, self).__init__(env, policy, rollouts_per_meta_task, meta_batch_size, max_path_length,
                                                envs_per_task=rollouts_per_meta_task, parallel=parallel)
        self.host_fallbacks = 0

    def obtain_samples(self, log=False, log_prefix=''):
        M, B, T = self.meta_batch_size, self.envs_per_task, self.max_path_length
        O, A = self.policy.obs_dim, self.policy.action_dim
        sess = self.policy.session
        ctx = sess.ensure(M * B * T, M * B)
        if sess.task_thetas is not None:           # parameters set while no context existed yet
            ctx.set_task_thetas(sess.task_thetas)
            sess.task_thetas = None
        slot = sess.next_slot()
        ctx.begin_rollout(slot, B, T)
        seed = int(np.random.randint(0, 2 ** 31 - 1))
        rewards = np.zeros((M * B, T), dtype=np.float32)
        infos = _StepTables(M * B, T)
        policy_seconds = env_seconds = 0.0
        observations = self.vec_env.reset()
        for t in range(T):
            started = time.time()
            actions = ctx.policy_step(slot, t, np.asarray(observations, dtype=np.float32).reshape(M, B, O), seed=seed,
                                      clip_infos=self.policy._pre_update_mode).reshape(M * B, A)
            policy_seconds += time.time() - started
            started = time.time()
            observations, step_rewards, finished, env_infos = self.vec_env.step(actions)
            env_seconds += time.time() - started
            rewards[:, t] = step_rewards
            infos.put_infos('env_infos', env_infos)
            infos.advance()
            if t < T - 1 and np.any(finished):      # ragged paths do not fit the fixed-length slab: collect on the host
                self.host_fallbacks += 1
                sess.step_cursor -= 1
                return super(DeviceSlabSampler, self).obtain_samples(log=log, log_prefix=log_prefix)
        ctx.set_rewards(slot, rewards.reshape(-1))
        sess._upload_counter += 1
        sess.upload_serial[slot] = sess._upload_counter
        slab = ctx.download_step(slot)
        paths = DevicePaths()
        for i in range(M):
            paths[i] = []
            log_std = np.tile(slab['old_log_std'][i], (T, 1))
            for b in range(B):
                env = i * B + b
                rows = slice(env * T, (env + 1) * T)
                path, _ = infos.cut(env)
                path.update(observations=slab['obs'][rows], actions=slab['act'][rows], rewards=slab['rew'][rows],
                            agent_infos=dict(mean=slab['old_mean'][rows], log_std=log_std))
                paths[i].append(path)
        paths.device_ref = (sess.serial, sess.upload_serial[slot], slot)
        paths.flat = dict(task_path_offsets=np.arange(M + 1, dtype=np.int32) * B,
                          path_row_offsets=np.arange(M * B + 1, dtype=np.int32) * T)
        self.total_timesteps_sampled += M * B * T
        if log:
            logger.logkv(log_prefix + 'PolicyExecTime', policy_seconds)
            logger.logkv(log_prefix + 'EnvExecTime', env_seconds)
        return paths
