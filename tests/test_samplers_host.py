"""Host-side rollout collection: environment pools and MetaSampler (contract of reference samplers/meta_sampler.py:59-137
and samplers/vectorized_env_executor.py:7-177; the reference's own checks are tests/test_samplers.py:132-189)."""
import numpy as np

from promp_amd.samplers.meta_sampler import MetaSampler
from promp_amd.samplers.vectorized_env_executor import MetaIterativeEnvExecutor, MetaParallelEnvExecutor


class CounterEnv(object):
    """deterministic toy: obs = [task, episode index, t]; terminates by itself after `task + 2` steps when task is odd"""

    def __init__(self):
        self.task, self.episode, self.t = 0, -1, 0

    def sample_tasks(self, n):
        return list(range(n))

    def set_task(self, task):
        self.task = task

    def reset(self):
        self.episode += 1
        self.t = 0
        return np.array([self.task, self.episode, self.t], dtype=np.float32)

    def step(self, action):
        self.t += 1
        done = bool(self.task % 2 == 1 and self.t >= self.task + 2)
        info = dict(t=self.t, nested=dict(twice=2.0 * self.t))
        return np.array([self.task, self.episode, self.t], dtype=np.float32), float(np.sum(action)) + self.t, done, info


class EchoPolicy(object):
    """actions = [obs[2], task index]; agent_infos carry mean / log_std like the Gaussian policy"""

    def __init__(self, M):
        self.M = M

    def get_actions(self, observations):
        assert len(observations) == self.M
        actions = [np.stack([o[:, 2], np.full(len(o), i)], axis=1).astype(np.float32) for i, o in enumerate(observations)]
        infos = [[dict(mean=a, log_std=np.zeros(2, np.float32)) for a in acts] for acts in actions]
        return actions, infos


def _drive(pool, n_steps, M, per):
    pool.set_tasks(list(range(M)))
    trace = [np.asarray(pool.reset())]
    for s in range(n_steps):
        obs, rew, done, infos = pool.step([np.array([s, e], np.float32) for e in range(M * per)])
        trace.append((np.asarray(obs), np.asarray(rew), np.asarray(done), [i['nested']['twice'] for i in infos]))
    return trace


def test_iterative_pool_resets_on_done_and_on_the_horizon():
    M, per, T = 3, 2, 4
    pool = MetaIterativeEnvExecutor(CounterEnv(), M, per, T)
    assert pool.num_envs == M * per
    tr = _drive(pool, 9, M, per)
    np.testing.assert_array_equal(tr[0][:, 0], [0, 0, 1, 1, 2, 2])            # task k -> environments [k per, (k+1) per)
    for s, (obs, rew, done, _) in enumerate(tr[1:], start=1):
        # even tasks never finish by themselves: the horizon (T = 4) ends their episodes; task 1 finishes after 3 steps
        np.testing.assert_array_equal(done[[0, 1, 4, 5]], [s % T == 0] * 4)
        np.testing.assert_array_equal(done[[2, 3]], [s % 3 == 0] * 2)
        # a finished environment hands back the first observation of its next episode
        assert all(obs[e, 2] == 0 for e in np.flatnonzero(done))
        assert all(obs[e, 2] > 0 for e in np.flatnonzero(~done))


def test_parallel_pool_matches_iterative_pool():
    M, per, T = 3, 2, 4
    a = _drive(MetaIterativeEnvExecutor(CounterEnv(), M, per, T), 7, M, per)
    pool = MetaParallelEnvExecutor(CounterEnv(), M, per, T)
    try:
        b = _drive(pool, 7, M, per)
    finally:
        pool.close()
    np.testing.assert_array_equal(a[0], b[0])
    for x, y in zip(a[1:], b[1:]):
        for u, v in zip(x, y):
            np.testing.assert_array_equal(np.asarray(u), np.asarray(v))


def test_meta_sampler_files_finished_trajectories_per_task():
    M, P, T = 4, 3, 5
    sampler = MetaSampler(CounterEnv(), EchoPolicy(M), rollouts_per_meta_task=P, meta_batch_size=M, max_path_length=T)
    sampler.update_tasks()
    paths = sampler.obtain_samples()
    assert list(paths.keys()) == list(range(M))
    total = 0
    for task, plist in paths.items():
        assert len(plist) >= P
        for p in plist:
            assert set(p.keys()) == {'observations', 'actions', 'rewards', 'env_infos', 'agent_infos'}
            n = len(p['rewards'])
            total += n
            assert n == (task + 2 if task % 2 == 1 and task + 2 < T else T)            # own termination or the horizon
            assert p['observations'].shape == (n, 3) and p['actions'].shape == (n, 2)
            np.testing.assert_array_equal(p['observations'][:, 0], task)               # filed under its own task
            np.testing.assert_array_equal(p['observations'][:, 2], np.arange(n))       # one episode, in order
            np.testing.assert_array_equal(p['actions'][:, 0], np.arange(n))
            np.testing.assert_array_equal(p['env_infos']['t'], np.arange(1, n + 1))
            np.testing.assert_array_equal(p['env_infos']['nested']['twice'], 2.0 * np.arange(1, n + 1))
            np.testing.assert_array_equal(p['agent_infos']['mean'], p['actions'])
            assert p['agent_infos']['log_std'].shape == (n, 2)
            np.testing.assert_allclose(p['rewards'], p['actions'].sum(axis=1) + np.arange(1, n + 1))
    assert total >= sampler.total_samples == M * P * T
    assert sampler.total_timesteps_sampled == M * P * T


class _ScriptedPolicy(object):
    """the same fixed action for every environment: the point leaves the start radius after a few steps"""

    def __init__(self, M, action):
        self.M, self.action = M, np.asarray(action, dtype=np.float64)

    def get_actions(self, observations):
        actions = [np.tile(self.action, (len(o), 1)) for o in observations]
        infos = [[dict(mean=a, log_std=np.zeros(2)) for a in acts] for acts in actions]
        return actions, infos


def test_sparse_point_rewards_keep_their_fraction():
    """normalize(MetaPointEnvCorner('sparse')) returns the Python int 0 inside the start radius and float progress rewards
    outside it; the step tables must promote like np.asarray over the reference's per-step lists
    (samplers/meta_sampler.py:100-125), not freeze the first column's int dtype (round-2 advisor finding)."""
    from promp_amd.envs.normalized_env import normalize
    from promp_amd.envs.point_env import MetaPointEnvCorner
    M, per, T = 2, 2, 12
    goal = np.array([2.0, 2.0])
    sampler = MetaSampler(normalize(MetaPointEnvCorner(reward_type='sparse')), _ScriptedPolicy(M, [10.0, 10.0]), per, M, T)
    sampler.vec_env.set_tasks([goal] * M)
    np.random.seed(11)
    paths = sampler.obtain_samples()
    # the same episodes stepped one environment at a time from each path's first observation
    ref_env = normalize(MetaPointEnvCorner(reward_type='sparse'))
    ref_env.set_task(goal)
    n_checked = 0
    for task in range(M):
        for path in paths[task]:
            assert path['rewards'].dtype == np.float64
            assert path['rewards'][0] == 0 and np.any((path['rewards'] != 0) & (np.abs(path['rewards']) < 0.29))
            ref_env.wrapped_env._state = np.array(path['observations'][0], dtype=np.float64)
            rewards = [ref_env.step(action)[1] for action in path['actions']]
            np.testing.assert_array_equal(path['rewards'], np.asarray(rewards, dtype=np.float64))
            n_checked += 1
    assert n_checked == M * per


# ---- lazily downloaded per-row results (promp_amd._lib.LazyRows / LazyResults, DevicePaths' pending side effect) ------------
class _FakeCtx(object):
    """stands in for _lib.Context: counts downloads, hands out recognisable data, runs the real pre-write protocol"""

    def __init__(self, n):
        import weakref
        from promp_amd import _lib
        self._lib, self._h, self.lib = _lib, True, None
        self.step_rows, self._lazy, self.calls, self.gen = {0: n}, {}, [], 0
        self._weakset = weakref.WeakSet

    def _call(self, name, step, returns, adv, *rest):
        self.calls.append(name)
        n = self.step_rows[step]
        np.ctypeslib.as_array(adv, shape=(n,))[:] = np.arange(n, dtype=np.float32) + 1000.0 * self.gen

    def download_raw(self, step):
        self.calls.append('promp_download_raw')
        n = self.step_rows[step]
        return np.arange(n, dtype=np.float64) * 2.0 + 1000.0 * self.gen, np.arange(n, dtype=np.float64) * 3.0 + 1000.0 * self.gen

    lazy_results = None
    _pre_write = None


def _fake_ctx(n, monkeypatch):
    from promp_amd import _lib
    ctx = _FakeCtx(n)
    ctx.lazy_results = lambda step: _lib.Context.lazy_results(ctx, step)
    ctx._pre_write = lambda step: _lib.Context._pre_write(ctx, step)
    monkeypatch.setattr(_lib.host_pool, 'get', lambda shape, dtype, lib=None: np.empty(shape, dtype))
    return ctx


def test_lazy_rows_behave_like_the_arrays_they_stand_for(monkeypatch):
    from promp_amd import _lib
    ctx = _fake_ctx(10, monkeypatch)
    res = ctx.lazy_results(0)
    adv, ret = _lib.LazyRows(res, 'advantages', 2, 7), _lib.LazyRows(res, 'returns', 0, 10)
    # what is known without crossing PCIe
    assert adv.shape == (5,) and len(adv) == 5 and adv.dtype == np.float32 and ret.dtype == np.float64 and adv.ndim == 1 and adv.size == 5
    assert not res.fetched and ctx.calls == []
    # first use fetches ONCE for all fields and all slices
    np.testing.assert_array_equal(np.asarray(adv), np.arange(2, 7, dtype=np.float32))
    assert res.fetched and ctx.calls == ['promp_download_processed', 'promp_download_raw']
    np.testing.assert_array_equal(ret, 2.0 * np.arange(10))
    assert ctx.calls == ['promp_download_processed', 'promp_download_raw']
    # arithmetic, reductions, indexing, conversions, concatenation
    assert (adv * 2.0).dtype == np.float32 and float((adv - 2.0).sum()) == 10.0 and float(np.mean(ret)) == 9.0 and adv[1] == 3.0
    assert np.asarray(adv, dtype=np.float64).dtype == np.float64 and adv.astype(np.float64)[-1] == 6.0 and adv.mean() == 4.0
    np.testing.assert_array_equal(np.concatenate([adv, adv]), np.tile(np.arange(2, 7, dtype=np.float32), 2))
    assert list(adv) == [2.0, 3.0, 4.0, 5.0, 6.0] and bool((adv == np.arange(2, 7)).all())
    out = np.zeros(5, np.float32)
    np.add(adv, 1.0, out=out)
    assert out[0] == 3.0


def test_lazy_results_are_brought_home_before_their_step_is_overwritten(monkeypatch):
    from promp_amd import _lib
    ctx = _fake_ctx(6, monkeypatch)
    kept, dropped = ctx.lazy_results(0), ctx.lazy_results(0)
    rows = _lib.LazyRows(kept, 'returns', 0, 6)
    del dropped                                   # nobody holds it any more: it must cost nothing
    before = _lib.LazyResults.fetch_count
    ctx._pre_write(0)                             # what upload_step / process_samples / set_advantages ... do first
    assert _lib.LazyResults.fetch_count == before + 1 and kept.fetched
    ctx.gen = 1                                   # the step now holds another batch
    np.testing.assert_array_equal(rows, 2.0 * np.arange(6))      # ... the rows handed out still show the first one
    fresh = ctx.lazy_results(0)
    np.testing.assert_array_equal(_lib.LazyRows(fresh, 'returns', 0, 6), 2.0 * np.arange(6) + 1000.0)
    ctx._pre_write(0)                             # nothing left to fetch
    assert _lib.LazyResults.fetch_count == before + 2


def test_device_paths_settle_the_pending_side_effect_on_first_look():
    from promp_amd.samplers.device_point_sampler import DevicePaths
    paths = DevicePaths([(0, [dict(rewards=np.zeros(3))]), (1, [dict(rewards=np.zeros(2))])])
    looked = []

    def side_effect():
        looked.append(1)
        for plist in paths.raw_values():
            for p in plist:
                p['returns'] = np.ones(len(p['rewards']))
    paths._pending = side_effect
    assert list(paths.keys()) == [0, 1] and len(paths) == 2 and not looked            # keys / len do not settle
    assert 'returns' not in list(paths.raw_values())[0][0]
    assert 'returns' in paths[1][0] and looked == [1]                                  # looking at a path list does, once
    assert all('returns' in p for plist in paths.values() for p in plist) and looked == [1]
    assert [k for k, _ in paths.items()] == [0, 1]


class FixedHorizonEnv(CounterEnv):
    """never terminates by itself (HalfCheetah-like): every episode is max_path_length long"""

    def step(self, action):
        obs, rew, _, info = CounterEnv.step(self, action)
        return obs, np.float32(rew), False, info


def test_fixed_horizon_batches_come_back_as_views_of_the_flat_arrays():
    """the sampler writes every finished episode once, into its final rows of the meta-batch's flat arrays; the path
    dicts are views of them and flatten_paths has nothing left to do -- field by field what the general route builds"""
    from collections import OrderedDict
    from promp_amd import _lib
    from promp_amd.samplers.meta_sampler import HostPaths
    M, B, T = 3, 4, 5
    sampler = MetaSampler(FixedHorizonEnv(), EchoPolicy(M), rollouts_per_meta_task=B, meta_batch_size=M, max_path_length=T, envs_per_task=2)
    sampler.update_tasks()
    paths = sampler.obtain_samples()
    assert isinstance(paths, HostPaths) and paths.flat is not None
    plain = OrderedDict((i, [dict(observations=p['observations'].copy(), actions=p['actions'].copy(), rewards=p['rewards'].copy(),
                                  env_infos=p['env_infos'], agent_infos={k: v.copy() for k, v in p['agent_infos'].items()})
                             for p in plist]) for i, plist in paths.items())
    fast, general = _lib.flatten_paths(paths), _lib.flatten_paths(plain)
    assert fast is paths.flat
    for key in general:
        np.testing.assert_array_equal(fast[key], general[key], err_msg=key)
        assert fast[key].dtype == general[key].dtype, key
    first = paths[0][0]
    assert np.shares_memory(first['observations'], fast['obs']) and np.shares_memory(paths[M - 1][-1]['agent_infos']['mean'], fast['old_mean'])
    # in-place edits are seen (same memory) ...
    first['rewards'][0] = 123.0
    assert _lib.flatten_paths(paths)['rew'][0] == 123.0
    # ... and a path whose array was REPLACED, or a dropped path, sends the batch down the general route
    paths[1][0]['rewards'] = paths[1][0]['rewards'] + 1.0
    again = _lib.flatten_paths(paths)
    assert again is not paths.flat and again['rew'][B * T] == general['rew'][B * T] + 1.0
    paths[1][0]['rewards'] = fast['rew'][B * T:B * T + T]          # (an equal view is still another object)
    paths2 = sampler.obtain_samples()
    paths2[2].pop()
    assert _lib.flatten_paths(paths2) is not paths2.flat


def test_copies_of_a_flat_backed_batch_take_the_general_route():
    """a deep copy or a pickle round trip keeps the identities flat_if_intact() compares but not the aliasing of the
    flat arrays; an in-place edit of the copy must reach the device"""
    import copy
    import pickle
    from promp_amd import _lib
    M, B, T = 3, 4, 5
    sampler = MetaSampler(FixedHorizonEnv(), EchoPolicy(M), rollouts_per_meta_task=B, meta_batch_size=M, max_path_length=T, envs_per_task=2)
    sampler.update_tasks()
    paths = sampler.obtain_samples()
    assert paths.flat_if_intact() is paths.flat
    for clone in (copy.deepcopy(paths), pickle.loads(pickle.dumps(paths)), copy.copy(paths)):
        assert type(clone) is type(paths) and list(clone.keys()) == list(paths.keys())
        assert clone.flat is None and clone.flat_if_intact() is None
        np.testing.assert_array_equal(clone[1][2]['rewards'], paths[1][2]['rewards'])
    clone = pickle.loads(pickle.dumps(paths))
    clone[0][0]['rewards'][0] = 999.0
    assert _lib.flatten_paths(clone)['rew'][0] == 999.0 and _lib.flatten_paths(paths)['rew'][0] != 999.0


def test_early_terminations_take_the_general_route_unchanged():
    M, B, T = 3, 2, 4
    sampler = MetaSampler(CounterEnv(), EchoPolicy(M), rollouts_per_meta_task=B, meta_batch_size=M, max_path_length=T)
    sampler.update_tasks()
    paths = sampler.obtain_samples()
    assert getattr(paths, 'flat', None) is None                     # task 1 ends its episodes after 3 steps and outgrows its share
    assert [len(p['rewards']) for p in paths[1]] == [3, 3, 3, 3] and all(len(p['rewards']) == 4 for p in paths[0])


def test_slab_backed_repoints_ordinary_path_dicts():
    from promp_amd import _lib, synthetic
    from promp_amd.samplers.meta_sampler import slab_backed
    rng = np.random.RandomState(0)
    theta = synthetic.init_theta(rng, 5, (8, 8), 2)
    paths = synthetic.make_paths(rng, theta, 3, 2, 7, 5, 2, (8, 8))
    general = _lib.flatten_paths(paths)
    general = {k: (v.copy() if v is not None else None) for k, v in general.items()}
    hp = slab_backed(paths)
    fl = _lib.flatten_paths(hp)
    assert fl is hp.flat
    for key in general:
        np.testing.assert_array_equal(fl[key], general[key], err_msg=key)
