"""MetaEnv: the interface MetaSampler, the vectorised executors and Trainer expect of a meta-environment (reference:
meta_policy_search/envs/base.py:6-49, a gym.Env subclass).  No gym dependency here: an environment is anything with reset() /
step(action) -> (obs, reward, done, info) plus the four methods below; the reference's own environments (its MuJoCo and Sawyer
classes are out of this build's scope) subclass its MetaEnv and satisfy this one as they are -- `MetaSampler(env=...)`
duck-types.  Subclass it to get the same NotImplementedError messages for a missing method.  (RandomEnv, base.py:51-149, the
MuJoCo parameter randomisation, has no counterpart: it needs mujoco_py.)"""


class MetaEnv(object):
    def sample_tasks(self, n_tasks):
        """-> a list of n_tasks tasks of the meta-environment (base.py:11-21)"""
        raise NotImplementedError

    def set_task(self, task):
        """make `task` the one the environment simulates (base.py:23-30)"""
        raise NotImplementedError

    def get_task(self):
        """-> the task the environment simulates (base.py:32-39)"""
        raise NotImplementedError

    def log_diagnostics(self, paths, prefix):
        """environment-specific logging over the paths of one iteration; nothing by default (base.py:41-49)"""
        pass

    # gym.Env's surface, for environments written against this class alone
    def reset(self):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError
