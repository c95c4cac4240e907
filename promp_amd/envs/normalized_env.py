"""normalize(env): the wrapper the reference's run scripts put around every environment
(meta_policy_search/envs/normalized_env.py:7-123).

Contract:
  * the policy acts in the box [-normalization_scale, normalization_scale]^A (default +-10); an action a is mapped affinely
    onto the wrapped environment's own action box [lb, ub]:  lb + (a + scale) (ub - lb) / (2 scale), then clipped to it;
  * optionally observations and / or rewards are normalised by exponentially weighted running estimates
    (obs: (o - mean) / (sqrt(var) + 1e-8); reward: r / (sqrt(var) + 1e-8)), updated on every sample with step sizes
    obs_alpha / reward_alpha;
  * everything else (set_task, sample_tasks, log_diagnostics, ...) is the wrapped environment's.
"""
import numpy as np


class _RunningMoments(object):
    """exponentially weighted mean / variance, one update per sample"""

    def __init__(self, shape, alpha):
        self.alpha = alpha
        self.mean = np.zeros(shape)
        self.var = np.ones(shape)

    def update(self, x):
        self.mean = (1 - self.alpha) * self.mean + self.alpha * x
        self.var = (1 - self.alpha) * self.var + self.alpha * np.square(x - self.mean)


class NormalizedEnv(object):
    def __init__(self, env, scale_reward=1., normalize_obs=False, normalize_reward=False, obs_alpha=0.001,
                 reward_alpha=0.001, normalization_scale=10.):
        self._wrapped_env = env
        self._scale_reward = 1            # (the reference ignores its scale_reward argument too: normalized_env.py:34)
        self._normalize_obs, self._normalize_reward = normalize_obs, normalize_reward
        self._obs = _RunningMoments(np.shape(env.observation_space.low) if hasattr(env, 'observation_space') else (), obs_alpha)
        self._rew = _RunningMoments((), reward_alpha)
        self._normalization_scale = normalization_scale

    @property
    def wrapped_env(self):
        return self._wrapped_env

    @property
    def observation_space(self):
        return self._wrapped_env.observation_space

    @property
    def action_space(self):
        box = self._wrapped_env.action_space
        if hasattr(box, 'low') and hasattr(box, 'high'):
            return type(box)(-self._normalization_scale, self._normalization_scale, np.shape(box.low))
        return box

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        return getattr(self._wrapped_env, name)

    def action_to_env(self, action):
        """policy-scale action -> the wrapped environment's action box"""
        box = self._wrapped_env.action_space
        if not (hasattr(box, 'low') and hasattr(box, 'high')):
            return action
        s = self._normalization_scale
        return np.clip(box.low + (np.asarray(action) + s) * (box.high - box.low) / (2 * s), box.low, box.high)

    def _see_obs(self, obs):
        if not self._normalize_obs:
            return obs
        self._obs.update(obs)
        return (obs - self._obs.mean) / (np.sqrt(self._obs.var) + 1e-8)

    def reset(self):
        return self._see_obs(self._wrapped_env.reset())

    def step(self, action):
        obs, reward, done, info = self._wrapped_env.step(self.action_to_env(action))
        if self._normalize_reward:
            self._rew.update(reward)
            reward = reward / (np.sqrt(self._rew.var) + 1e-8)
        return self._see_obs(obs), reward * self._scale_reward, done, info

    def __getstate__(self):
        return self.__dict__

    def __setstate__(self, state):
        self.__dict__.update(state)


normalize = NormalizedEnv
