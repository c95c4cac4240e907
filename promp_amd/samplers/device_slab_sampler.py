"""MetaSampler for host-stepped environments (MuJoCo ...) whose policy queries fill the sampling step's slab ON THE DEVICE
(SURVEY.md 8f row 1).

The reference pays, per environment step, one sess.run with the observations fed and the means fetched, host-side noise,
and at the end of the rollout a re-upload of everything as feed dicts (policies/meta_gaussian_mlp_policy.py:99-157,
samplers/meta_sampler.py:87-125, meta_algos/base.py:245-301).  Here a step is one small call:

    actions = ctx.policy_step(slot, s, observations)      # [M, B, O] in, [M, B, A] out

the device evaluates every task's mean network, draws the exploration noise (Philox4x32-10, keyed by a seed taken from
NumPy's RNG once per sampling step) and keeps observation, action and mean of vectorised step s.  Episodes may end whenever
the environment says so: like the reference's sampler (meta_sampler.py:100-125) an environment that reports `done` starts
its next episode at once, collection stops when meta_batch_size * rollouts_per_meta_task * max_path_length steps belong to
FINISHED episodes, and episodes still running at that point are dropped.  Because the slab (task x path x t rows) cannot be
laid out before the lengths are known, the rows wait in a staging area under (s, environment) and promp_end_collection copies
the finished episodes into path order on the device; the rewards follow in the same call.
MetaSampleProcessor.process_samples then finds the data resident: no trajectory upload at all.  The path dicts of the plugin
API are materialised from one download per sampling step.
"""
import time

import numpy as np

from ..utils import logger
from .device_point_sampler import DevicePaths
from .meta_sampler import MetaSampler, _StepTables


class DeviceSlabSampler(MetaSampler):
    def __init__(self, env, policy, rollouts_per_meta_task, meta_batch_size, max_path_length, envs_per_task=None,
                 parallel=False):
        super(DeviceSlabSampler, self).__init__(env, policy, rollouts_per_meta_task, meta_batch_size, max_path_length,
                                                envs_per_task=envs_per_task, parallel=parallel)
        self.host_fallbacks = 0        # (kept for callers that logged it: the device path serves every episode structure now)

    def obtain_samples(self, log=False, log_prefix=''):
        M, B, T = self.meta_batch_size, self.envs_per_task, self.max_path_length
        O, A = self.policy.obs_dim, self.policy.action_dim
        n_envs = M * B
        # every environment finishes an episode at least every T steps, so after S steps at least n_envs (S - T + 1) steps are
        # in finished episodes: 2T - 1 steps always reach total_samples (with envs_per_task below rollouts_per_meta_task: more)
        max_steps = T - 1 + -(-self.total_samples // n_envs)
        max_rows = max_steps * n_envs
        sess = self.policy.session
        ctx = sess.ensure(max_rows, max_rows)
        if sess.task_thetas is not None:           # parameters set while no context existed yet
            ctx.set_task_thetas(sess.task_thetas)
            sess.task_thetas = None
        slot = sess.next_slot()
        ctx.begin_collection(slot, B, max_steps)
        seed = int(np.random.randint(0, 2 ** 31 - 1))
        rewards = np.zeros((max_steps, n_envs), dtype=np.float32)
        infos = _StepTables(n_envs, T)
        started_at = np.zeros(n_envs, dtype=np.int64)          # vectorised step at which every environment's episode began
        done_paths = [[] for _ in range(M)]                    # per task, in the reference's order: (env, start, length, env_infos)
        policy_seconds = env_seconds = 0.0
        collected, s = 0, 0
        observations = self.vec_env.reset()
        while collected < self.total_samples:
            assert s < max_steps, 'the environment pool let an episode run past max_path_length'
            started = time.time()
            actions = ctx.policy_step(slot, s, np.asarray(observations, dtype=np.float32).reshape(M, B, O), seed=seed,
                                      clip_infos=self.policy._pre_update_mode).reshape(n_envs, A)
            policy_seconds += time.time() - started
            started = time.time()
            observations, step_rewards, finished, env_infos = self.vec_env.step(actions)
            env_seconds += time.time() - started
            rewards[s] = step_rewards
            infos.put_infos('env_infos', env_infos)
            infos.advance()
            for env in np.flatnonzero(np.asarray(finished)):
                path, n = infos.cut(env)
                done_paths[env // B].append((env, int(started_at[env]), n, path.get('env_infos', {})))
                collected += n
                started_at[env] = s + 1
            s += 1
        empty = [i for i, tp in enumerate(done_paths) if not tp]
        if empty:
            # (the reference's sampler would hand such a task an empty path list and fail later, in the sample processor;
            #  typical cause: envs_per_task above rollouts_per_meta_task, so total_samples is reached before every task's
            #  environments have finished an episode)
            raise RuntimeError('DeviceSlabSampler: no finished episode for task(s) %s within total_samples = %d '
                               '(envs_per_task = %d, rollouts_per_meta_task = %d, max_path_length = %d)'
                               % (empty, self.total_samples, B, self.batch_size, T))
        flat = [p for task_paths in done_paths for p in task_paths]
        tpo = np.concatenate([[0], np.cumsum([len(tp) for tp in done_paths])]).astype(np.int32)
        env_of = np.array([p[0] for p in flat], dtype=np.int32)
        start_of = np.array([p[1] for p in flat], dtype=np.int32)
        len_of = np.array([p[2] for p in flat], dtype=np.int32)
        ctx.end_collection(slot, tpo, env_of, start_of, len_of,
                           np.concatenate([rewards[a:a + n, e] for e, a, n in zip(env_of, start_of, len_of)]))
        sess._upload_counter += 1
        sess.upload_serial[slot] = sess._upload_counter
        slab = ctx.download_step(slot)
        pro = np.concatenate([[0], np.cumsum(len_of)]).astype(np.int32)
        paths = DevicePaths()
        for i in range(M):
            paths[i] = []
            for p in range(tpo[i], tpo[i + 1]):
                rows = slice(pro[p], pro[p + 1])
                paths[i].append(dict(observations=slab['obs'][rows], actions=slab['act'][rows], rewards=slab['rew'][rows],
                                     env_infos=flat[p][3],
                                     agent_infos=dict(mean=slab['old_mean'][rows], log_std=np.tile(slab['old_log_std'][i], (len_of[p], 1)))))
        paths.device_ref = (sess.serial, sess.upload_serial[slot], slot)
        # (the downloaded slab arrays ride along: the sample processor hands out per-task views of them instead of concatenating
        #  the path dicts' slices again)
        paths.flat = dict(task_path_offsets=tpo, path_row_offsets=pro, path_env=env_of, path_start=start_of,
                          obs=slab['obs'], act=slab['act'], rew=slab['rew'], old_mean=slab['old_mean'], old_log_std=slab['old_log_std'])
        self.total_timesteps_sampled += self.total_samples
        if log:
            logger.logkv(log_prefix + 'PolicyExecTime', policy_seconds)
            logger.logkv(log_prefix + 'EnvExecTime', env_seconds)
        return paths
