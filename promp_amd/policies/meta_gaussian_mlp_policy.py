"""MetaGaussianMLPPolicy: the parameter container the algorithm reads and writes
(reference: policies/meta_gaussian_mlp_policy.py:9-157, policies/gaussian_mlp_policy.py:31-123,
policies/base.py:25-34,173-286).

Parameters live on the GPU as one flat float32 vector in the reference's OrderedDict order; get_param_values /
set_params / policies_params_vals / update_task_parameters expose them under the reference's names
('mean_network/hidden_0/kernel', ..., 'log_std_network/log_std_var').

get_actions (rollout-side inference, SURVEY.md 8f "next" row 1) evaluates the mean network on the device
(promp_policy_forward) under every task's current parameters; the Gaussian exploration noise is drawn on the host
with NumPy's RNG (the reference draws it with TensorFlow's, so rollouts were never bit-reproducible across runs).
"""
from collections import OrderedDict

import numpy as np

from .. import session as session_mod
from ..utils import logger
from .distributions.diagonal_gaussian import DiagonalGaussian


class MetaGaussianMLPPolicy(object):
    def __init__(self, meta_batch_size, obs_dim, action_dim, name='policy', hidden_sizes=(32, 32), learn_std=True,
                 hidden_nonlinearity='tanh', output_nonlinearity=None, init_std=1., min_std=1e-6,
                 n_tasks_global=None, rank=None, world=None, device_id=None, **kwargs):
        """meta_batch_size is the number of tasks THIS process holds.  In a task-sharded run (one process per GPU under
        torchrun) rank / world / device default to RANK / WORLD_SIZE / LOCAL_RANK and n_tasks_global to
        meta_batch_size * world: the meta-gradient is then the mean over all ranks' tasks (one RCCL all-reduce per epoch)."""
        # policies/base.py:31, networks/mlp.py:47: tanh (the default), relu, or None = LINEAR hidden layers (what
        # tf.layers.dense(activation=None) builds) -- anything else is refused by name, nothing is silently replaced
        from .._lib import hidden_act_id, HIDDEN_ACTS, output_act_id, OUTPUT_ACTS
        act_id = hidden_act_id(hidden_nonlinearity)
        self.hidden_nonlinearity = [k for k, v in HIDDEN_ACTS.items() if v == act_id][0]
        # output_nonlinearity (mlp.py:53-60, 114-117): applied to the mean network's last layer; None (every run script), tanh or relu
        out_id = output_act_id(output_nonlinearity)
        self.output_nonlinearity = [k for k, v in OUTPUT_ACTS.items() if v == out_id][0]
        self.meta_batch_size = int(meta_batch_size)
        self.obs_dim, self.action_dim = int(np.prod(obs_dim)), int(np.prod(action_dim))
        self.name = name
        self.hidden_sizes = tuple(hidden_sizes)
        self.learn_std = learn_std
        self.min_log_std, self.init_log_std = np.log(min_std), np.log(init_std)
        self._dist = DiagonalGaussian(self.action_dim)
        sizes = (self.obs_dim,) + self.hidden_sizes + (self.action_dim,)
        self._shapes = OrderedDict()
        for i in range(len(sizes) - 1):
            lname = 'output' if i == len(sizes) - 2 else 'hidden_%d' % i
            self._shapes['mean_network/%s/kernel' % lname] = (sizes[i], sizes[i + 1])
            self._shapes['mean_network/%s/bias' % lname] = (sizes[i + 1],)
        self._shapes['log_std_network/log_std_var'] = (1, self.action_dim)
        self.policy_params_keys = list(self._shapes.keys())
        # xavier-uniform kernels, zero biases, log_std = log(init_std)   (mlp.py:12-13, gaussian_mlp_policy.py:63-69)
        parts = []
        for k, shp in self._shapes.items():
            if k.endswith('kernel'):
                lim = np.sqrt(6.0 / (shp[0] + shp[1]))
                parts.append(np.random.uniform(-lim, lim, size=shp))
            elif k.endswith('bias'):
                parts.append(np.zeros(shp))
            else:
                parts.append(np.full(shp, self.init_log_std))
        from .. import comm
        env_rank, env_world, env_local = comm.env_world()
        rank = env_rank if rank is None else int(rank)
        world = env_world if world is None else int(world)
        device_id = (env_local if world > 1 else 0) if device_id is None else int(device_id)
        self.session = session_mod.DeviceSession(self.meta_batch_size, self.obs_dim, self.action_dim, self.hidden_sizes,
                                                 n_tasks_global=n_tasks_global or self.meta_batch_size * world,
                                                 device_id=device_id, rank=rank, world=world, hidden_act=self.hidden_nonlinearity,
                                                 output_act=self.output_nonlinearity)
        self.session.min_std = float(min_std)
        self.session.learn_std = bool(learn_std)      # False: log_std is neither adapted nor trained (gaussian_mlp_policy.py:63-69)
        self.session.set_theta(self._flatten(parts))
        self._log_std_cache = None                    # ([M, A] raw log_std of the tasks' current parameters, parameter version)
        self._pre_update_mode = True
        self.switch_to_pre_update()

    # ---- flat <-> named ----
    def _flatten(self, parts):
        return np.concatenate([np.asarray(p, dtype=np.float32).reshape(-1) for p in parts])

    def _unflatten(self, theta):
        out, off = OrderedDict(), 0
        for k, shp in self._shapes.items():
            n = int(np.prod(shp))
            out[k] = theta[off:off + n].reshape(shp).copy()
            off += n
        return out

    @property
    def distribution(self):
        return self._dist

    # ---- reference API: parameters ----
    def get_params(self):
        return self.get_param_values()

    def get_param_values(self):
        return self._unflatten(self.session.get_theta())

    def set_params(self, policy_params):
        assert all(k1 == k2 for k1, k2 in zip(self.policy_params_keys, policy_params.keys())), \
            'parameter keys must match with variable'
        self.session.set_theta(self._flatten(list(policy_params.values())))
        self.session.param_version += 1

    def switch_to_pre_update(self):
        """policies/base.py:234-240: get_action uses the pre-update policy; per-task params = theta replicated"""
        self._pre_update_mode = True
        s = self.session
        s.step_cursor = 0
        s.param_version += 1
        if s.ctx is not None:
            s.ctx.switch_to_pre_update()
            s.task_thetas = None
        else:
            s.task_thetas = np.tile(s.theta, (self.meta_batch_size, 1))

    def _task_thetas(self):
        s = self.session
        return s.ctx.get_task_thetas() if s.ctx is not None else s.task_thetas

    @property
    def policies_params_vals(self):
        return [self._unflatten(t) for t in self._task_thetas()]

    def update_task_parameters(self, updated_policies_parameters):
        """policies/base.py:262-269"""
        th = np.stack([self._flatten(list(d.values())) for d in updated_policies_parameters])
        s = self.session
        s.task_thetas = th
        s.param_version += 1
        if s.ctx is not None:
            s.ctx.set_task_thetas(th)
        self._pre_update_mode = False

    # ---- reference API: acting ----
    def get_actions(self, observations):
        """observations: list[M] of [B,O] -> (list[M] of [B,A], list[M] of list[B] of {mean, log_std})
        (meta_gaussian_mlp_policy.py:99-157)"""
        assert len(observations) == self.meta_batch_size
        batch_size = observations[0].shape[0]
        assert all(obs.shape[0] == batch_size for obs in observations)
        s = self.session
        ctx = s.ensure()
        if s.task_thetas is not None:          # parameters set while no context existed yet
            ctx.set_task_thetas(s.task_thetas)
            s.task_thetas = None
        means = ctx.policy_forward(np.stack([np.asarray(o, dtype=np.float32).reshape(batch_size, -1) for o in observations]))
        raws = self._log_std_tasks()
        actions, agent_infos = [], []
        for i in range(self.meta_batch_size):
            raw = raws[i]
            log_std = np.maximum(raw, self.min_log_std) if self._pre_update_mode else raw   # gaussian_mlp_policy.py:71 vs :182
            actions.append(means[i] + np.random.normal(size=means[i].shape) * np.exp(raw))   # noise uses the raw variable (:74)
            agent_infos.append([dict(mean=m, log_std=log_std) for m in means[i]])
        return actions, agent_infos

    def _log_std_tasks(self):
        """[M, A] raw log_std of every task's current parameters.  Cached per parameter version: the sampler calls
        get_actions once per environment step, and a download of all [M, Theta] parameters per step would dominate the
        rollout (the version is bumped by everything that changes the tasks' parameters)."""
        ver = self.session.param_version
        if self._log_std_cache is None or self._log_std_cache[1] != ver:
            self._log_std_cache = (np.array(self._task_thetas()[:, -self.action_dim:], dtype=np.float32), ver)
        return self._log_std_cache[0]

    def get_action(self, observation, task=0):
        obs = [np.expand_dims(observation, 0)] * self.meta_batch_size
        a, infos = self.get_actions(obs)
        return a[task][0], infos[task][0]

    def reset(self, dones=None):
        pass

    def log_diagnostics(self, paths, prefix=''):
        """gaussian_mlp_policy.py:118-123"""
        log_stds = np.vstack([p['agent_infos']['log_std'] for p in paths])
        logger.logkv(prefix + 'AveragePolicyStd', np.mean(np.exp(log_stds)))
