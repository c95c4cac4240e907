"""promp_amd.utils.utils / envs.base / samplers.utils: the host-side helpers run scripts and user code import beside the plugin
classes (reference: meta_policy_search/utils/utils.py:43-190, envs/base.py:6-49, samplers/utils.py:5-71).  Checked against the
definitions themselves (scipy's lfilter for the discounted sum, as the reference computes it) -- no GPU, no library."""
import json

import numpy as np
import pytest

from promp_amd.envs.base import MetaEnv
from promp_amd.samplers.utils import rollout
from promp_amd.utils import utils as U


def test_discount_cumsum_is_the_reference_filter():
    scipy_signal = pytest.importorskip('scipy.signal')
    rng = np.random.RandomState(0)
    for x in (rng.randn(200), rng.randn(1), rng.randn(37, 3), rng.randn(50).astype(np.float32)):
        for g in (0.99, 1.0, 0.0, 0.5):
            ref = scipy_signal.lfilter([1], [1, float(-g)], x[::-1], axis=0)[::-1]       # utils.py:81
            np.testing.assert_allclose(U.discount_cumsum(x, g), ref, rtol=1e-12, atol=1e-12)
    assert U.discount_cumsum(np.ones(3), 0.5).tolist() == [1.75, 1.5, 1.0]


def test_advantage_helpers_and_explained_variance():
    rng = np.random.RandomState(1)
    a = rng.randn(100) * 3 + 2
    n = U.normalize_advantages(a)
    assert abs(n.mean()) < 1e-12 and abs(n.std() - 1) < 1e-6
    p = U.shift_advantages_to_positive(a)
    assert p.min() == 1e-8 and np.allclose(p - p.min(), a - a.min())
    y = rng.randn(50)
    assert U.explained_variance_1d(y, y) == pytest.approx(1.0, abs=1e-6)
    assert U.explained_variance_1d(np.zeros(50), y) == pytest.approx(0.0, abs=1e-6)
    assert U.explained_variance_1d(np.zeros(5), np.ones(5)) == 1          # constant target, constant prediction
    assert U.explained_variance_1d(np.arange(5.0), np.ones(5)) == 0       # constant target, varying prediction
    with pytest.raises(AssertionError):
        U.explained_variance_1d(np.zeros((2, 2)), np.zeros((2, 2)))


def test_dict_list_helpers():
    steps = [dict(a=np.array([1.0, 2.0]), info=dict(k=np.array([3]))), dict(a=np.array([4.0, 5.0]), info=dict(k=np.array([6])))]
    c = U.concat_tensor_dict_list(steps)
    assert c['a'].tolist() == [1.0, 2.0, 4.0, 5.0] and c['info']['k'].tolist() == [3, 6]
    s = U.stack_tensor_dict_list(steps)
    assert s['a'].shape == (2, 2) and s['info']['k'].shape == (2, 1)
    assert U.extract(dict(x=1, y=2, z=3), 'x', 'z') == (1, 3)
    assert U.extract([dict(x=1, y=2), dict(x=3, y=4)], 'y', 'x') == ([2, 4], [1, 3])
    with pytest.raises(NotImplementedError):
        U.extract(3, 'x')


def test_set_seed_and_class_encoder(capsys):
    U.set_seed(7)
    a = np.random.rand(3)
    U.set_seed(7 + 4294967294)            # taken modulo 2^32 - 2 like the reference (utils.py:172)
    assert np.array_equal(a, np.random.rand(3)) and 'using seed 7' in capsys.readouterr().out
    cfg = dict(env=MetaEnv, fn=np.tanh, lr=1e-3)
    d = json.loads(json.dumps(cfg, cls=U.ClassEncoder))
    assert d == {'env': {'$class': 'promp_amd.envs.base.MetaEnv'}, 'fn': {'function': 'tanh'}, 'lr': 0.001}
    with pytest.raises(TypeError):
        json.dumps(dict(x=object()), cls=U.ClassEncoder)


def test_meta_env_interface_and_rollout():
    env = MetaEnv()
    for call in (lambda: env.sample_tasks(2), lambda: env.set_task(0), env.get_task, env.reset, lambda: env.step(0)):
        with pytest.raises(NotImplementedError):
            call()
    assert env.log_diagnostics([], 'p') is None

    class Space(object):
        def __init__(self, n): self.shape = (n,)

    class Walk(MetaEnv):
        observation_space, action_space = Space(2), Space(2)
        def reset(self):
            self.s, self.t = np.zeros(2), 0
            return self.s.copy()
        def step(self, a):
            self.s, self.t = self.s + a, self.t + 1
            return self.s.copy(), float(-np.abs(self.s).sum()), self.t >= 4, dict(t=self.t)

    class Agent(object):
        resets = 0
        def reset(self): self.resets += 1
        def get_action(self, o): return np.array([[0.5, -0.25]]), dict(mean=np.zeros(2))

    agent = Agent()
    p = rollout(Walk(), agent, max_path_length=10)
    assert agent.resets == 1 and len(p['rewards']) == 4 and p['actons'] is p['actions']          # done after four steps
    assert np.allclose(p['observations'][3], [1.5, -0.75]) and p['actions'][0].shape == (2,) and p['env_infos'][3] == dict(t=4)
    assert len(rollout(Walk(), agent, max_path_length=3)['rewards']) == 3
    assert len(rollout(Walk(), agent, max_path_length=6, ignore_done=True)['rewards']) == 6
    with pytest.raises(NotImplementedError):
        rollout(Walk(), agent, animated=True)
