"""2-D point-mass meta-environment with corner goals: the environment of BASELINE config 1
(run_scripts/pro-mp_run_point_mass.py trains on normalize(MetaPointEnvCorner())).

Behaviour (reference: meta_policy_search/envs/point_envs/point_env_2d_corner.py:13-93, MetaEnv interface envs/base.py:6-49):
  * a task is a goal, one of the corners (-2,-2), (2,-2), (-2,2), (2,2), drawn uniformly;
  * reset: state ~ U(-0.2, 0.2)^2; step: state += clip(action, -0.2, 0.2); an episode never ends by itself;
  * reward_type 'dense': -|s' - goal|;  'dense_squared': -|s' - goal|^2;
    'sparse' (the default): nothing while the point is within L1 distance `sparse_reward_radius` of the origin, afterwards the
    progress |s - goal| - |s' - goal| if the goal is the corner nearest to s', else 0.
The device runs the same arithmetic in k_point_rollout (samplers/device_point_sampler.py); the trajectories of the reference
class are the golden vectors tests/golden/point_env_*.npz.
"""
import numpy as np

from .base import MetaEnv


class ActionBox(object):
    """just the attributes the normalize wrapper and the run scripts read from a gym.spaces.Box"""

    def __init__(self, low, high, shape):
        self.low = np.full(shape, low, dtype=np.float64)
        self.high = np.full(shape, high, dtype=np.float64)
        self.shape = tuple(shape)


class MetaPointEnvCorner(MetaEnv):
    CORNERS = np.array([[-2.0, -2.0], [2.0, -2.0], [-2.0, 2.0], [2.0, 2.0]])

    def __init__(self, reward_type='sparse', sparse_reward_radius=0.5):
        assert reward_type in ('dense', 'dense_squared', 'sparse')
        self.reward_type = reward_type
        self.sparse_reward_radius = sparse_reward_radius
        self.corners = [c.copy() for c in self.CORNERS]
        self.observation_space = ActionBox(-np.inf, np.inf, (2,))
        self.action_space = ActionBox(-0.2, 0.2, (2,))
        self.goal = self.corners[3]
        self._state = np.zeros(2)

    # ---- MetaEnv ----
    def sample_tasks(self, n_tasks):
        return [self.corners[k] for k in np.random.choice(len(self.corners), size=n_tasks)]

    def set_task(self, task):
        self.goal = np.asarray(task, dtype=np.float64)

    def get_task(self):
        return self.goal

    def log_diagnostics(self, *args, **kwargs):
        pass

    # ---- dynamics ----
    def reset(self):
        self._state = np.random.uniform(-0.2, 0.2, size=(2,))
        return self._state.copy()

    def step(self, action):
        before = self._state
        self._state = before + np.clip(action, self.action_space.low, self.action_space.high)
        return self._state.copy(), self.reward(before, action, self._state), False, {}

    @staticmethod
    def _distance(a, b):
        d = np.asarray(a, dtype=np.float64) - b
        return np.sqrt(np.sum(d * d))          # (sum of squares, then root: the rounding of a row-wise 2-norm)

    def reward(self, obs, act, obs_next):
        to_goal = self._distance(obs_next, self.goal)
        if self.reward_type == 'dense':
            return -to_goal
        if self.reward_type == 'dense_squared':
            return -to_goal ** 2
        if np.sum(np.abs(obs_next)) < self.sparse_reward_radius:
            return 0
        if to_goal == min(self._distance(obs_next, corner) for corner in self.corners):
            return self._distance(obs, self.goal) - to_goal
        return 0


MetaPointEnv = MetaPointEnvCorner      # (the name earlier revisions of this package used)
