"""A NumPy 2-D point environment with corner goals -- the shape of BASELINE config 0
(run_scripts/pro-mp_run_point_mass.py: Point2D, obs 2, act 2).  Written from the MetaEnv interface
(reference: meta_policy_search/envs/base.py:6-49); used by the end-to-end trainer test."""
import numpy as np


class MetaPointEnv(object):
    def __init__(self, reward_type='dense'):
        self.goal = np.array([2.0, 2.0])
        self.state = np.zeros(2)
        self.reward_type = reward_type

    def sample_tasks(self, n_tasks):
        corners = np.array([[-2, -2], [-2, 2], [2, -2], [2, 2]], dtype=np.float64)
        return [corners[i] for i in np.random.choice(4, n_tasks)]

    def set_task(self, task):
        self.goal = np.asarray(task, dtype=np.float64)

    def get_task(self):
        return self.goal

    def reset(self):
        self.state = np.random.uniform(-0.2, 0.2, size=2)
        return self.state.copy()

    def step(self, action):
        self.state = self.state + np.clip(action, -0.1, 0.1)
        dist = np.linalg.norm(self.state - self.goal)
        return self.state.copy(), -dist, False, {'goal_dist': dist}

    def log_diagnostics(self, paths, prefix=''):
        pass
