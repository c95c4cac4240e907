"""The drop-in boundary: the reference's plugin classes re-created over the HIP library.
CPU: runs against the kernel emulator at tiny shapes.  The same scenarios run on the GPU in test_gpu_plugin_api.py."""
import os
from collections import OrderedDict

import numpy as np
import pytest

from oracle import sample_processing as sp
from promp_amd import _lib, session
from tests import devlib, helpers


@pytest.fixture(autouse=True)
def _two_emulated_cus(monkeypatch):
    monkeypatch.setenv('PROMP_EMU_CUS', '2')   # API-level tests: fewer host threads per emulated launch


@pytest.fixture()
def emu():
    lib = devlib.emu_library()
    _lib.set_library_for_testing(lib)
    yield lib
    _lib.set_library_for_testing(None)
    session._current = None


def run_processor_scenario():
    from promp_amd.baselines.linear_baseline import LinearFeatureBaseline
    from promp_amd.samplers.meta_sample_processor import MetaSampleProcessor
    meta, paths, g = helpers.load_sample_proc('ragged')
    proc = MetaSampleProcessor(baseline=LinearFeatureBaseline(), **meta['kwargs'])
    with pytest.raises(AssertionError):
        proc.process_samples(list(paths.values()))            # "paths must be a dict" (meta_sample_processor.py:25)
    out = proc.process_samples(paths, log=False)
    assert isinstance(out, list) and len(out) == len(paths)
    # reference tests/test_samplers.py:172-189: 8 keys, advantages.size == N
    assert set(out[0].keys()) == {'observations', 'actions', 'rewards', 'returns', 'advantages', 'env_infos',
                                  'agent_infos', 'adj_avg_rewards'}
    assert out[0]['advantages'].size == out[0]['rewards'].size
    cat = lambda k: np.concatenate([sd[k] for sd in out])
    np.testing.assert_allclose(cat('returns'), g['returns'], rtol=1e-12)          # float64, as the reference
    np.testing.assert_allclose(cat('advantages'), g['advantages'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(cat('adj_avg_rewards'), g['adj_avg_rewards'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(proc.baseline.get_param_values(), g['coeffs'][-1], rtol=1e-5, atol=1e-7)
    # side effect on the path dicts (samplers/base.py:104,159): raw (un-normalised) GAE advantages
    ref_paths = helpers.load_sample_proc('ragged')[1]
    sp.process_samples_meta(ref_paths, baseline_kind=sp.BASELINE_LINEAR_FEATURE, **meta['kwargs'])
    np.testing.assert_allclose(paths[1][2]['advantages'], ref_paths[1][2]['advantages'], rtol=1e-7, atol=1e-9)


def run_algo_scenario(M=2, P=2, T=30, O=5, A=3, hidden=(32, 32), K=1, epochs=2):
    """policy + sample processor + ProMP wired like run_scripts/pro-mp_run_mujoco.py:21-77, on synthetic paths."""
    from oracle import policy as op, promp as pm
    from promp_amd import synthetic
    from promp_amd.baselines.linear_baseline import LinearFeatureBaseline
    from promp_amd.meta_algos.pro_mp import ProMP
    from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
    from promp_amd.samplers.meta_sample_processor import MetaSampleProcessor
    np.random.seed(3)
    policy = MetaGaussianMLPPolicy(name='meta-policy', obs_dim=O, action_dim=A, meta_batch_size=M, hidden_sizes=hidden)
    proc = MetaSampleProcessor(baseline=LinearFeatureBaseline(), discount=0.99, gae_lambda=1, normalize_adv=True)
    algo = ProMP(policy=policy, inner_lr=0.1, meta_batch_size=M, num_inner_grad_steps=K, learning_rate=1e-3,
                 num_ppo_steps=epochs, clip_eps=0.3, target_inner_step=0.01, init_inner_kl_penalty=5e-4,
                 adaptive_inner_kl_penalty=False)
    params0 = policy.get_param_values()
    assert list(params0.keys())[0] == 'mean_network/hidden_0/kernel' and params0['log_std_network/log_std_var'].shape == (1, A)
    theta0 = np.concatenate([v.reshape(-1) for v in params0.values()])
    spec = op.PolicySpec(O, A, hidden)
    rng = np.random.RandomState(5)
    policy.switch_to_pre_update()
    all_samples, all_paths = [], []
    for step in range(K + 1):
        th = np.stack([np.concatenate([v.reshape(-1) for v in d.values()]) for d in policy.policies_params_vals])
        paths = synthetic.make_paths(rng, th, M, P, T, O, A, hidden)
        all_paths.append(paths)
        sd = proc.process_samples(paths, log=False)
        all_samples.append(sd)
        if step < K:
            algo._adapt(sd)
    # oracle on the same inputs
    kw = dict(baseline_kind=sp.BASELINE_LINEAR_FEATURE, discount=0.99, gae_lambda=1, normalize_adv=True)
    ref_samples = [sp.process_samples_meta(p, **kw)[0] for p in all_paths]
    ad = pm.adapt(spec, [theta0.astype(np.float64)] * M, ref_samples[0], np.full(spec.n_params, 0.1))
    th1 = np.stack([np.concatenate([v.reshape(-1) for v in d.values()]) for d in policy.policies_params_vals])
    if K == 1:
        assert np.max(np.abs(th1 - np.stack(ad))) < 1e-5
    algo.optimize_policy(all_samples, log=False)
    th_ref, ref = pm.optimize_policy(spec, theta0.astype(np.float64), ref_samples, np.full(spec.n_params, 0.1),
                                     np.full(K, 5e-4), 0.3, pm.AdamState(spec.n_params), 1e-3, epochs)
    np.testing.assert_allclose(algo.last_stats['loss_before'], ref['loss_before'], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(algo.last_stats['loss_after'], ref['loss_after'], rtol=1e-3, atol=1e-5)
    th_new = np.concatenate([v.reshape(-1) for v in policy.get_param_values().values()])
    assert np.mean(np.abs((th_new - theta0) - (th_ref - theta0))) < 0.05 * np.mean(np.abs(th_ref - theta0))
    # samples that did NOT come from our processor (plain dicts) are uploaded on the fly
    plain = [[dict(sd) for sd in step] for step in all_samples]
    policy.set_params(params0)
    algo2_stats_before = algo.last_stats['loss_before']
    algo.session.ctx.set_adam_state(np.zeros(spec.n_params), np.zeros(spec.n_params), 0)
    algo.optimize_policy(plain, log=False)
    np.testing.assert_allclose(algo.last_stats['loss_before'], algo2_stats_before, rtol=1e-5, atol=1e-6)


def run_trainer_scenario(n_itr=2, device_rollouts=False, log_dir=None, resume=None, build_only=False):
    from promp_amd.baselines.linear_baseline import LinearFeatureBaseline
    from promp_amd.envs.normalized_env import normalize
    from promp_amd.envs.point_env import MetaPointEnvCorner
    from promp_amd.meta_algos.pro_mp import ProMP
    from promp_amd.meta_trainer import Trainer
    from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
    from promp_amd.samplers.meta_sample_processor import MetaSampleProcessor
    from promp_amd.samplers.meta_sampler import MetaSampler
    from promp_amd.utils import logger
    logger.configure(dir=log_dir, snapshot_mode='last', quiet=True)
    np.random.seed(1)
    M, P, T = 2, 3, 12
    env = normalize(MetaPointEnvCorner(reward_type='dense'))       # run_scripts/pro-mp_run_point_mass.py:27-28
    policy = MetaGaussianMLPPolicy(name='meta-policy', obs_dim=2, action_dim=2, meta_batch_size=M, hidden_sizes=(32, 32))
    from promp_amd.samplers.device_point_sampler import DevicePointEnvSampler
    sampler = (DevicePointEnvSampler if device_rollouts else MetaSampler)(
        env=env, policy=policy, rollouts_per_meta_task=P, meta_batch_size=M, max_path_length=T, parallel=False)
    proc = MetaSampleProcessor(baseline=LinearFeatureBaseline(), discount=0.99, gae_lambda=1, normalize_adv=True)
    algo = ProMP(policy=policy, inner_lr=0.1, meta_batch_size=M, num_inner_grad_steps=1, learning_rate=1e-3, num_ppo_steps=2,
                 clip_eps=0.3, init_inner_kl_penalty=5e-4, adaptive_inner_kl_penalty=False)
    trainer = Trainer(algo=algo, policy=policy, env=env, sampler=sampler, sample_processor=proc, n_itr=n_itr, num_inner_grad_steps=1)
    if resume:
        trainer.load_snapshot(resume)
    if build_only:
        return trainer
    before = policy.get_param_values()
    seen = {}
    orig = logger.dumpkvs

    def grab():
        seen.update(logger.getkvs())
        return orig()
    logger.dumpkvs = grab
    try:
        trainer.train()
    finally:
        logger.dumpkvs = orig
    # the reference's logging keys (meta_trainer.py:131-142, samplers/base.py:143-149, pro_mp.py:195-199)
    for k in ['Itr', 'n_timesteps', 'Time-OuterStep', 'Time-TotalInner', 'Time-InnerStep', 'Time-SampleProc', 'Time-Sampling',
              'Time', 'ItrTime', 'Time-MAMLSteps', 'Step_0-AverageReturn', 'Step_1-AverageDiscountedReturn', 'Step_0-NumTrajs',
              'Step_1-StdReturn', 'Step_0-MaxReturn', 'Step_0-MinReturn', 'Step_0-AveragePolicyStd', 'LossBefore', 'LossAfter',
              'KLInner', 'KLCoeffInner']:
        assert k in seen, k
    assert seen['n_timesteps'] == n_itr * 2 * M * P * T
    after = policy.get_param_values()
    assert any(np.any(before[k] != after[k]) for k in before) and all(np.all(np.isfinite(v)) for v in after.values())


def run_device_rollout_scenario(M=3, B=4, T=15, hidden=(32, 32), reward_type='sparse', hidden_act='tanh'):
    """DevicePointEnvSampler: the trajectories equal a float64 NumPy rollout of the same environment with the same start
    states and noise (oracle/point_rollout.py, itself pinned by the reference environment's own trajectories); process_samples
    takes the resident slab (no upload) and returns what it returns for the same paths handed over as plain host dicts."""
    from oracle import policy as op, point_rollout as pr
    from promp_amd.baselines.linear_baseline import LinearFeatureBaseline
    from promp_amd.envs.normalized_env import normalize
    from promp_amd.envs.point_env import MetaPointEnvCorner
    from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
    from promp_amd.samplers.device_point_sampler import DevicePointEnvSampler
    from promp_amd.samplers.meta_sample_processor import MetaSampleProcessor
    np.random.seed(21)
    env = normalize(MetaPointEnvCorner(reward_type=reward_type))
    policy = MetaGaussianMLPPolicy(name='p', obs_dim=2, action_dim=2, meta_batch_size=M, hidden_sizes=hidden,
                                   hidden_nonlinearity=None if hidden_act == 'identity' else hidden_act)
    sampler = DevicePointEnvSampler(env=env, policy=policy, rollouts_per_meta_task=B, meta_batch_size=M, max_path_length=T)
    sampler.update_tasks()
    spec = op.PolicySpec(2, 2, hidden, hidden_act=hidden_act)
    theta = spec.from_ordered_dict(policy.get_param_values())
    theta[-2:] = np.log(12.0)             # wide exploration: actions beyond the wrapper's +-10 box, points that leave the start region
    policy.set_params(spec.to_ordered_dict(theta))
    policy.switch_to_pre_update()
    state = np.random.get_state()
    paths = sampler.obtain_samples()
    np.random.set_state(state)                                   # replay the sampler's draws for the oracle
    start = np.random.uniform(-0.2, 0.2, size=(M, B, 2))
    noise = np.random.normal(size=(M, B, T, 2)).astype(np.float32)
    ref = pr.rollout(spec, np.tile(theta, (M, 1)), sampler.goals, start, noise, clip_infos=True, reward_type=reward_type,
                     normalization_scale=10.0, max_step=0.2, sparse_radius=0.5)
    assert list(paths.keys()) == list(range(M)) and all(len(paths[i]) == B for i in range(M))
    cat = lambda key, sub=None: np.concatenate([(p[key] if sub is None else p[key][sub]) for i in range(M) for p in paths[i]])
    np.testing.assert_allclose(cat('observations'), ref['obs'], atol=2e-6)
    np.testing.assert_allclose(cat('actions'), ref['act'], atol=2e-6)
    np.testing.assert_allclose(cat('rewards'), ref['rew'], atol=2e-6)
    if reward_type == 'sparse':
        assert 0 < np.count_nonzero(ref['rew']) < ref['rew'].size
    np.testing.assert_allclose(cat('agent_infos', 'mean'), ref['mean'], atol=2e-6)
    np.testing.assert_allclose(paths[0][0]['agent_infos']['log_std'][0], ref['log_std'][0], atol=1e-6)
    assert sampler.total_timesteps_sampled == M * B * T
    # processing: resident (no upload) == uploaded
    proc = MetaSampleProcessor(baseline=LinearFeatureBaseline(), discount=0.99, gae_lambda=1, normalize_adv=True)
    assert proc.lazy_host_arrays is False      # the default is the reference's behaviour: ndarrays, path dicts written in the call
    sess = policy.session
    serial_before = list(sess.upload_serial)
    held = list(paths.raw_values())[0]
    sd_eager = proc.process_samples(paths)
    assert isinstance(sd_eager[0]['returns'], np.ndarray) and isinstance(held[0]['returns'], np.ndarray), \
        'references to the path dicts taken before the call see the side effect (samplers/base.py:104,159)'
    import pickle
    proc.lazy_host_arrays = True               # opt-in
    sd_dev = proc.process_samples(paths)
    assert sess.upload_serial == serial_before, 'device paths must not be uploaded again'
    # a resident batch gets its per-row results back lazily: nothing has crossed PCIe yet ...
    assert isinstance(sd_dev[0]['returns'], _lib.LazyRows) and isinstance(sd_dev[0]['advantages'], _lib.LazyRows)
    assert sd_dev[0]['advantages'].shape == sd_dev[0]['rewards'].shape and sd_dev[0]['advantages'].dtype == np.float32
    fetched = _lib.LazyResults.fetch_count
    raw = list(paths.raw_values())            # (not paths[i]: looking at the paths settles the pending side effect)
    host_paths = OrderedDict((i, [dict(observations=p['observations'], actions=p['actions'], rewards=p['rewards'],
                                        env_infos=p['env_infos'], agent_infos=p['agent_infos']) for p in raw[i]]) for i in range(M))
    assert _lib.LazyResults.fetch_count == fetched
    # ... and the upload of another batch into the session first brings home what was handed out (one fetch), so the lazy
    # arrays still hold the FIRST batch's results afterwards
    sd_host = proc.process_samples(host_paths)
    for a, b in zip(sd_dev, sd_host):
        for key in ('observations', 'actions', 'rewards', 'returns', 'advantages', 'adj_avg_rewards'):
            np.testing.assert_allclose(a[key], b[key], rtol=1e-6, atol=1e-7)
    assert _lib.LazyResults.fetch_count == fetched + 1
    unpickled = pickle.loads(pickle.dumps(sd_dev[1]['returns']))       # a LazyRows pickles as the rows themselves
    assert isinstance(unpickled, np.ndarray) and np.array_equal(unpickled, np.asarray(sd_eager[1]['returns']))
    # the per-path side effect (samplers/base.py:104,159) is settled when the paths are looked at; eager == lazy, value for value
    ret_lazy = np.concatenate([np.asarray(p['returns']) for i in range(M) for p in paths[i]])
    adv_lazy = np.concatenate([np.asarray(p['advantages']) for i in range(M) for p in paths[i]])
    ret_host = np.concatenate([p['returns'] for i in range(M) for p in host_paths[i]])
    adv_host = np.concatenate([p['advantages'] for i in range(M) for p in host_paths[i]])
    np.testing.assert_allclose(ret_lazy, ret_host, rtol=1e-12)
    np.testing.assert_allclose(adv_lazy, adv_host, rtol=1e-6, atol=1e-9)
    assert (sd_dev[0]['advantages'] * 2.0).dtype == np.float32 and float(np.mean(sd_dev[0]['returns'])) == float(sd_host[0]['returns'].mean())
    # steady state of a training loop: samples released before the next batch is processed -> no download at all
    del sd_dev, sd_host
    fetched = _lib.LazyResults.fetch_count
    for _ in range(2):
        sd = proc.process_samples(paths)
        assert sd[0].device_ref is not None
        del sd
    assert _lib.LazyResults.fetch_count == fetched
    # the opt-out hands out plain arrays
    proc.lazy_host_arrays = False
    sd = proc.process_samples(paths)
    assert isinstance(sd[0]['advantages'], np.ndarray) and isinstance(OrderedDict.__getitem__(paths, 0)[0]['returns'], np.ndarray)


def test_device_rollout_point_env(emu):
    run_device_rollout_scenario(reward_type='sparse')
    run_device_rollout_scenario(M=2, B=3, T=9, reward_type='dense_squared')


def run_policy_step_scenario(M=2, B=3, T=5, O=4, A=3, hidden=(32, 32), hidden_act='tanh', output_act=None):
    """DeviceSlabSampler: every environment step is one promp_policy_step; the slab rows equal the oracle's forward pass plus
    the oracle's Philox noise; rewards arrive once; process_samples uploads nothing; an early `done` falls back to the host."""
    from oracle import philox, policy as op
    from promp_amd.baselines.linear_baseline import LinearFeatureBaseline
    from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
    from promp_amd.samplers.device_slab_sampler import DeviceSlabSampler
    from promp_amd.samplers.meta_sample_processor import MetaSampleProcessor

    class DriftEnv(object):
        """obs = running sum of clipped actions (first O entries cycled); reward = -|obs|; never done unless told"""
        def __init__(self, stop_at=None):
            self.stop_at, self.t, self.s, self.task = stop_at, 0, np.zeros(O), 0.0
        def sample_tasks(self, n): return list(np.arange(n, dtype=np.float64))
        def set_task(self, task): self.task = float(task)
        def reset(self):
            self.t, self.s = 0, np.full(O, 0.1 * self.task)
            return self.s.copy()
        def step(self, a):
            self.t += 1
            self.s = self.s + 0.05 * np.resize(np.clip(a, -1, 1), O)
            return self.s.copy(), -float(np.abs(self.s).sum()), bool(self.stop_at and self.t >= self.stop_at), dict(t=self.t)

    np.random.seed(5)
    policy = MetaGaussianMLPPolicy(name='p', obs_dim=O, action_dim=A, meta_batch_size=M, hidden_sizes=hidden,
                                   hidden_nonlinearity=None if hidden_act == 'identity' else hidden_act, output_nonlinearity=output_act)
    sampler = DeviceSlabSampler(env=DriftEnv(), policy=policy, rollouts_per_meta_task=B, meta_batch_size=M, max_path_length=T)
    sampler.update_tasks()
    policy.switch_to_pre_update()
    state = np.random.get_state()
    paths = sampler.obtain_samples()
    np.random.set_state(state)
    seed = int(np.random.randint(0, 2 ** 31 - 1))
    spec = op.PolicySpec(O, A, hidden, hidden_act=hidden_act, output_act=output_act or 'identity')
    theta = spec.from_ordered_dict(policy.get_param_values()).astype(np.float64)
    assert sampler.host_fallbacks == 0 and all(len(paths[i]) == B for i in range(M))
    cat = lambda key, sub=None: np.concatenate([(p[key] if sub is None else p[key][sub]) for i in range(M) for p in paths[i]])
    obs = cat('observations')
    mean = op.forward(spec, theta, obs.astype(np.float64), False)[0]
    # Philox counter = the staging row (vectorised step s, environment); sampling step 0 -> stream.  Fixed-length episodes: the
    # row of path (task i, env b) at time t is t * M * B + i * B + b
    counters = np.concatenate([np.arange(T) * (M * B) + env for env in range(M * B)])
    noise = philox.action_noise(seed, counters, A, stream=0)
    np.testing.assert_allclose(cat('agent_infos', 'mean'), mean, atol=2e-6)
    np.testing.assert_allclose(cat('actions'), mean + np.exp(theta[-A:]) * noise, atol=5e-6)
    act = cat('actions')
    nxt = obs.astype(np.float64) + 0.05 * np.stack([np.resize(np.clip(a, -1, 1), O) for a in act.astype(np.float64)])
    np.testing.assert_allclose(cat('rewards'), -np.abs(nxt).sum(axis=1), atol=2e-6)             # what the host uploaded at the end
    for i in range(M):
        for p in paths[i]:
            assert p['observations'].shape == (T, O) and p['actions'].shape == (T, A)
            np.testing.assert_array_equal(p['env_infos']['t'], np.arange(1, T + 1))
            np.testing.assert_allclose(p['observations'][0], 0.1 * i, atol=1e-7)             # filed under its own task
            np.testing.assert_allclose(p['observations'][1:], p['observations'][:-1] + 0.05 * np.resize(
                np.clip(p['actions'][:-1], -1, 1).T, (O, T - 1)).T, atol=1e-6)              # the env really got these actions
    proc = MetaSampleProcessor(baseline=LinearFeatureBaseline(), discount=0.99, gae_lambda=1, normalize_adv=True)
    before = list(policy.session.upload_serial)
    sd = proc.process_samples(paths)
    assert policy.session.upload_serial == before and len(sd) == M
    # ---- episodes that end early (meta_sampler.py:100-125): same device path, no host fallback.  An environment whose
    # trajectory does not depend on the actions makes the result comparable with the host-side MetaSampler field by field
    # (the two draw different exploration noise): same paths per task in the same order, same lengths, observations, rewards, infos
    class ClockEnv(object):
        """episode e of task k lasts 2 + (k + e) % 4 steps; observations / rewards are functions of (task, episode, t)"""
        def __init__(self):
            self.task, self.episode, self.t = 0, -1, 0
        def sample_tasks(self, n): return list(range(n))
        def set_task(self, task): self.task = task
        def log_diagnostics(self, *a, **k): pass
        def _obs(self):
            return np.array([self.task, self.episode, self.t] + [0.5] * (O - 3), dtype=np.float32)[:O]
        def reset(self):
            self.episode += 1
            self.t = 0
            return self._obs()
        def step(self, a):
            self.t += 1
            done = self.t >= 2 + (self.task + self.episode) % 4
            return self._obs(), float(10 * self.task + self.episode + 0.25 * self.t), done, dict(t=self.t)

    from promp_amd.samplers.meta_sampler import MetaSampler
    sampler2 = DeviceSlabSampler(env=ClockEnv(), policy=policy, rollouts_per_meta_task=B, meta_batch_size=M, max_path_length=T)
    host = MetaSampler(env=ClockEnv(), policy=policy, rollouts_per_meta_task=B, meta_batch_size=M, max_path_length=T)
    for smp in (sampler2, host):
        smp.update_tasks()
    policy.switch_to_pre_update()
    state = np.random.get_state()
    paths2 = sampler2.obtain_samples()
    np.random.set_state(state)
    seed2 = int(np.random.randint(0, 2 ** 31 - 1))
    ref2 = host.obtain_samples()
    assert sampler2.host_fallbacks == 0
    assert [len(paths2[i]) for i in range(M)] == [len(ref2[i]) for i in range(M)]
    lengths = set()
    for i in range(M):
        for p, q in zip(paths2[i], ref2[i]):
            lengths.add(len(p['rewards']))
            np.testing.assert_array_equal(p['observations'], np.asarray(q['observations'], dtype=np.float32))
            np.testing.assert_array_equal(p['rewards'], np.asarray(q['rewards'], dtype=np.float32))
            np.testing.assert_array_equal(p['env_infos']['t'], q['env_infos']['t'])
            assert p['actions'].shape == (len(p['rewards']), A) and p['agent_infos']['log_std'].shape == p['actions'].shape
    assert len(lengths) > 1 and max(lengths) < T                  # really ragged, really early
    # the rows of every path are the network's output on its observations plus the staging row's noise
    cat2 = lambda key, sub=None: np.concatenate([(p[key] if sub is None else p[key][sub]) for i in range(M) for p in paths2[i]])
    mean2 = op.forward(spec, theta, cat2('observations').astype(np.float64), False)[0]
    np.testing.assert_allclose(cat2('agent_infos', 'mean'), mean2, atol=2e-6)
    fl = paths2.flat
    counters2 = np.concatenate([(st + np.arange(n)) * (M * B) + e for e, st, n in
                                zip(fl['path_env'], fl['path_start'], np.diff(fl['path_row_offsets']))])
    noise2 = philox.action_noise(seed2, counters2, A, stream=policy.session.upload_serial.index(paths2.device_ref[1]))
    np.testing.assert_allclose(cat2('actions'), mean2 + np.exp(theta[-A:]) * noise2, atol=5e-6)
    sd2 = proc.process_samples(paths2)
    assert len(sd2) == M and policy.session.resident_slot(sd2) is not None       # processed where it was collected: no upload
    assert sum(len(d['rewards']) for d in sd2) == sum(len(p['rewards']) for i in range(M) for p in paths2[i])


def run_vpg_scenario(M=3, P=3, T=30, O=5, A=3, hidden=(32, 32), inner_type='log_likelihood', exploration=False):
    """VPGMAML.optimize_policy: one Adam step on the log-likelihood meta-objective (vpg_maml.py:65-175), against the oracle"""
    from oracle import policy as op, promp as pm, trpo as otrpo
    from promp_amd.meta_algos.vpg_maml import VPGMAML
    from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
    from promp_amd.utils import logger
    logger.configure(quiet=True)
    theta, all_slabs, _ = helpers.make_promp_case(55, M, P, T, O, A, hidden, 1)
    spec = op.PolicySpec(O, A, hidden)
    kind = 'loglik' if inner_type == 'log_likelihood' else 'ratio'
    policy = MetaGaussianMLPPolicy(name='p', obs_dim=O, action_dim=A, meta_batch_size=M, hidden_sizes=hidden)
    policy.set_params(spec.to_ordered_dict(theta))
    algo = VPGMAML(policy=policy, learning_rate=1e-3, inner_type=inner_type, inner_lr=0.1, meta_batch_size=M,
                   num_inner_grad_steps=1, exploration=exploration)
    samples = [[dict(observations=s['observations'], actions=s['actions'], advantages=s['advantages'],
                     agent_infos=s['agent_infos']) for s in step] for step in all_slabs]
    coeffs = None
    if exploration:
        rng = np.random.RandomState(56)
        for d in samples[-1]:
            d['adj_avg_rewards'] = rng.randn(len(d['advantages'])).astype(np.float32) + 0.5
        coeffs = np.array([np.mean(d['adj_avg_rewards']) for d in samples[-1]], np.float64)
    algo.optimize_policy(samples, log=False)
    t64, alpha = theta.astype(np.float64), np.full(spec.n_params, 0.1)

    def objective(th, grad):
        r = pm.meta_objective_and_grad(spec, th, all_slabs, alpha, np.zeros(1), 0.0, kind, 'loglik', want_grad=grad)
        v, g = (otrpo.exploration_term(spec, th, all_slabs[0], coeffs, grad) if exploration else (0.0, 0.0))
        return r['loss'] + v, (r['grad'] + g if grad else None)
    loss_before, g = objective(t64, True)
    th_ref = pm.adam_step(t64, g, pm.AdamState(spec.n_params), 1e-3)
    loss_after, _ = objective(th_ref, False)
    np.testing.assert_allclose(algo.last_stats['loss_before'], loss_before, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(algo.last_stats['loss_after'], loss_after, rtol=2e-4, atol=1e-6)
    got = spec.from_ordered_dict(policy.get_param_values())
    d_dev, d_ref = got - theta, th_ref - t64            # Adam's first step is lr * sign(g) wherever |g| >> eps
    assert np.mean(np.sign(d_dev) == np.sign(d_ref)) > 0.98 and np.max(np.abs(d_dev)) < 1.01e-3


def run_dice_processor_scenario(name='ragged'):
    """DiceMetaSampleProcessor against the reference's own outputs (tests/golden/dice_proc_*.npz)"""
    import json
    from promp_amd.baselines.linear_baseline import LinearFeatureBaseline, LinearTimeBaseline
    from promp_amd.baselines.zero_baseline import ZeroBaseline
    from promp_amd.samplers.dice_sample_processor import DiceMetaSampleProcessor
    g = np.load(os.path.join(helpers.GOLDEN, 'dice_proc_%s.npz' % name))
    meta = json.loads(str(g['meta']))
    paths = helpers.dice_paths_from_golden(g)
    kinds = dict(zero=ZeroBaseline, linear_feature=LinearFeatureBaseline, linear_time=LinearTimeBaseline)
    base = kinds[meta['baseline']]()
    extra = dict(return_baseline=kinds[meta['return_baseline']]()) if meta.get('return_baseline') else {}
    proc = DiceMetaSampleProcessor(base, max_path_length=meta['max_path_length'], **extra, **meta['kwargs'])
    with pytest.raises(AssertionError):
        proc.process_samples(list(paths.values()))
    out = proc.process_samples(paths, log=False)
    assert len(out) == len(paths)
    assert set(out[0].keys()) >= set(meta['keys'])                 # the reference's keys (mask, adjusted_rewards, ...)
    for i, sd in enumerate(out):
        np.testing.assert_array_equal(sd['mask'], g['mask'][i])
        np.testing.assert_array_equal(sd['rewards'], g['padded_rewards'][i])
        np.testing.assert_array_equal(sd['observations'], g['padded_observations'][i])
        np.testing.assert_allclose(sd['adjusted_rewards'], g['adjusted_rewards'][i], rtol=1e-5, atol=1e-6)
        if extra:      # return_baseline: GAE advantages beside the DiCE rewards (dice_sample_processor.py:113-124, 196-238)
            np.testing.assert_allclose(sd['advantages'], g['advantages'][i], rtol=1e-4, atol=1e-5)
    return proc, out


def run_vpg_dice_maml_scenario(name='k1_ragged'):
    """VPG_DICEMAML._adapt + optimize_policy (one Adam step on the exact meta-gradient) against the oracle / torch.autograd fixture"""
    from oracle import dice, policy as op, promp as pm
    from promp_amd.meta_algos.vpg_dice_maml import VPG_DICEMAML
    from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
    from promp_amd.utils import logger
    logger.configure(quiet=True)
    g = np.load(os.path.join(helpers.GOLDEN, 'vpgdice_autograd_%s.npz' % name))
    c, t64, all_slabs = helpers.dice_case_from_golden(g)
    spec = op.PolicySpec(c['O'], c['A'], c['hidden'])
    theta = t64.astype(np.float32)
    policy = MetaGaussianMLPPolicy(name='p', obs_dim=c['O'], action_dim=c['A'], meta_batch_size=c['M'], hidden_sizes=c['hidden'])
    policy.set_params(spec.to_ordered_dict(theta))
    algo = VPG_DICEMAML(c['Tmax'], policy=policy, learning_rate=1e-3, inner_lr=c['alpha'], meta_batch_size=c['M'],
                        num_inner_grad_steps=c['K'])
    assert algo.name == 'vpg_dice_maml' and 'advantages' in algo._optimization_keys
    samples = [[sl['padded'] for sl in step] for step in all_slabs]
    alpha = np.full(spec.n_params, c['alpha'])
    with pytest.raises(AssertionError):          # the last step's samples must carry the advantages
        algo.optimize_policy([[{k: v for k, v in sd.items() if k != 'advantages'} for sd in step] for step in samples], log=False)
    algo.optimize_policy(samples, log=False)
    r = dice.meta_objective_and_grad(spec, t64, all_slabs, alpha, outer='vpg')
    th_ref = pm.adam_step(t64, r['grad'], pm.AdamState(spec.n_params), 1e-3)
    np.testing.assert_allclose(algo.last_stats['loss_before'], r['loss'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(algo.last_stats['loss_before'], float(g['loss']), rtol=1e-4, atol=1e-6)
    after = dice.meta_objective_and_grad(spec, th_ref, all_slabs, alpha, want_grad=False, outer='vpg')
    np.testing.assert_allclose(algo.last_stats['loss_after'], after['loss'], rtol=5e-4, atol=1e-5)
    got = spec.from_ordered_dict(policy.get_param_values())
    d_dev, d_ref = got - theta, th_ref - t64
    assert np.mean(np.sign(d_dev) == np.sign(d_ref)) > 0.98 and np.max(np.abs(d_dev)) < 1.01e-3


def run_dice_maml_scenario(name='k1_ragged'):
    """DICEMAML._adapt + optimize_policy (one Adam step on the exact DiCE meta-gradient) against the oracle"""
    from oracle import dice, policy as op, promp as pm
    from promp_amd.meta_algos.dice_maml import DICEMAML
    from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
    from promp_amd.utils import logger
    logger.configure(quiet=True)
    g = np.load(os.path.join(helpers.GOLDEN, 'dice_autograd_%s.npz' % name))
    c, t64, all_slabs = helpers.dice_case_from_golden(g)
    spec = op.PolicySpec(c['O'], c['A'], c['hidden'])
    theta = t64.astype(np.float32)
    policy = MetaGaussianMLPPolicy(name='p', obs_dim=c['O'], action_dim=c['A'], meta_batch_size=c['M'], hidden_sizes=c['hidden'])
    policy.set_params(spec.to_ordered_dict(theta))
    algo = DICEMAML(c['Tmax'], policy=policy, learning_rate=1e-3, inner_lr=c['alpha'], meta_batch_size=c['M'],
                    num_inner_grad_steps=c['K'])
    samples = [[sl['padded'] for sl in step] for step in all_slabs]
    alpha = np.full(spec.n_params, c['alpha'])
    # inner step through the plugin API
    policy.switch_to_pre_update()
    algo._adapt(samples[0])
    ad = dice.adapt(spec, [t64] * c['M'], all_slabs[0], alpha)
    got = np.stack([spec.from_ordered_dict(d) for d in policy.policies_params_vals])
    assert np.max(np.abs((got - theta) - (np.stack(ad) - t64))) < 1e-4 * np.max(np.abs(np.stack(ad) - t64))
    # outer step
    algo.optimize_policy(samples, log=False)
    r = dice.meta_objective_and_grad(spec, t64, all_slabs, alpha)
    th_ref = pm.adam_step(t64, r['grad'], pm.AdamState(spec.n_params), 1e-3)
    np.testing.assert_allclose(algo.last_stats['loss_before'], r['loss'], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(algo.last_stats['loss_before'], float(g['loss']), rtol=1e-6, atol=1e-9)
    got = spec.from_ordered_dict(policy.get_param_values())
    d_dev, d_ref = got - theta, th_ref - t64
    assert np.mean(np.sign(d_dev) == np.sign(d_ref)) > 0.98 and np.max(np.abs(d_dev)) < 1.01e-3


def test_dice_sample_processor(emu):
    run_dice_processor_scenario('ragged')
    run_dice_processor_scenario('raw')


def test_dice_maml(emu):
    run_dice_maml_scenario('k1_ragged')


def test_dice_sample_processor_with_return_baseline(emu):
    run_dice_processor_scenario('retbase')


def test_vpg_dice_maml(emu):
    run_vpg_dice_maml_scenario('k1_ragged')


def test_vpg_maml(emu):
    run_vpg_scenario(inner_type='log_likelihood')
    run_vpg_scenario(M=2, P=2, T=20, inner_type='likelihood_ratio', exploration=True)


def test_logger_formats_and_snapshot_modes(tmp_path):
    from promp_amd.utils import logger
    d = str(tmp_path / 'run')
    logger.configure(dir=d, format_strs=['csv', 'json', 'log'], snapshot_mode='last_gap', snapshot_gap=2, quiet=True)
    for itr in range(4):
        logger.logkv('Itr', itr)
        logger.logkv('LossAfter', np.float32(0.5 - itr))
        if itr >= 2:
            logger.logkv('LateKey', 7 * itr)                  # a key that first appears later extends the csv header
        logger.log('iteration', itr)
        logger.save_itr_params(itr, dict(itr=itr, x=np.arange(3) + itr))
        logger.dumpkvs()
    import csv, json
    rows = list(csv.DictReader(open(os.path.join(d, 'progress.csv'))))
    assert [r['Itr'] for r in rows] == ['0', '1', '2', '3'] and rows[0]['LateKey'] == '' and rows[3]['LateKey'] == '21'
    js = [json.loads(l) for l in open(os.path.join(d, 'progress.json'))]
    assert len(js) == 4 and js[2]['LateKey'] == 14 and abs(js[1]['LossAfter'] + 0.5) < 1e-6
    assert 'iteration 3' in open(os.path.join(d, 'log.txt')).read()
    snap = logger.load_params(os.path.join(d, 'params.pkl'))     # last_gap: params.pkl only on multiples of the gap (0, 2)
    assert snap['itr'] == 2 and not any(f.startswith('itr_') for f in os.listdir(d))
    for mode, expect in (('all', {'itr_0.pkl', 'itr_1.pkl', 'itr_2.pkl'}), ('gap', {'itr_0.pkl', 'itr_2.pkl'}), ('last', {'params.pkl'}),
                         ('none', set())):
        dd = str(tmp_path / mode)
        logger.configure(dir=dd, format_strs=['csv'], snapshot_mode=mode, snapshot_gap=2, quiet=True)
        for itr in range(3):
            logger.save_itr_params(itr, dict(itr=itr))
        assert {f for f in os.listdir(dd) if f.endswith('.pkl')} == expect, mode
    with pytest.raises(NotImplementedError):
        logger.configure(dir=str(tmp_path / 'bad'), snapshot_mode='sometimes')
    logger.configure(quiet=True)


def test_trainer_snapshot_round_trip(emu, tmp_path):
    """a run resumed from its snapshot continues with the same parameters, Adam state and KL coefficients"""
    from promp_amd.utils import logger
    d = str(tmp_path / 'snap')
    run_trainer_scenario(n_itr=2, log_dir=d)
    snap = logger.load_params(os.path.join(d, 'params.pkl'))
    assert snap['itr'] == 1 and snap['adam_t'] == 2 * 2
    trainer = run_trainer_scenario(n_itr=3, resume=os.path.join(d, 'params.pkl'), build_only=True)
    assert trainer.start_itr == 2
    now = trainer.policy.get_param_values()
    for k, v in snap['policy_params'].items():
        np.testing.assert_array_equal(now[k], v)
    m, v, t = trainer.policy.session.ensure().get_adam_state()
    np.testing.assert_array_equal(m, snap['adam_m'])
    assert t == snap['adam_t']
    np.testing.assert_array_equal(trainer.algo.inner_kl_coeff, snap['inner_kl_coeff'])
    trainer.train()                                            # one more iteration (itr 2) runs from the restored state
    assert trainer.policy.session.ctx.get_adam_state()[2] == 3 * 2


def test_policy_step_fills_the_slab(emu):
    run_policy_step_scenario()


def test_rollout_kernels_on_any_layer_table(emu):
    """Shapes outside the two-layer kernels (a third layer, more than 8 actions, relu / linear hidden units) collect on the device too:
    one workgroup per environment walks the layer table (k_gen_policy_step / k_gen_point_rollout)."""
    run_policy_step_scenario(M=2, B=3, T=5, O=9, A=11, hidden=(48, 40, 24))
    run_policy_step_scenario(M=2, B=2, T=4, O=5, A=3, hidden=(20,), hidden_act='relu')
    run_policy_step_scenario(M=2, B=2, T=4, O=5, A=3, hidden=(32, 32), output_act='tanh')        # output_nonlinearity (mlp.py:114-117)
    run_device_rollout_scenario(M=2, B=3, T=9, hidden=(32, 16, 24), reward_type='sparse')
    run_device_rollout_scenario(M=2, B=2, T=6, hidden=(24, 24), reward_type='dense', hidden_act='identity')


def test_device_rollout_point_env_with_device_noise(emu):
    from oracle import philox, policy as op, point_rollout as pr
    from promp_amd.envs.normalized_env import normalize
    from promp_amd.envs.point_env import MetaPointEnvCorner
    from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
    from promp_amd.samplers.device_point_sampler import DevicePointEnvSampler
    M, B, T, hidden = 2, 3, 7, (32, 32)
    np.random.seed(8)
    policy = MetaGaussianMLPPolicy(name='p', obs_dim=2, action_dim=2, meta_batch_size=M, hidden_sizes=hidden)
    sampler = DevicePointEnvSampler(env=normalize(MetaPointEnvCorner(reward_type='dense')), policy=policy, rollouts_per_meta_task=B,
                                    meta_batch_size=M, max_path_length=T, device_noise=True)
    sampler.update_tasks()
    policy.switch_to_pre_update()
    state = np.random.get_state()
    paths = sampler.obtain_samples()
    np.random.set_state(state)
    start = np.random.uniform(-0.2, 0.2, size=(M, B, 2))
    seed = int(np.random.randint(0, 2 ** 31 - 1))
    noise = philox.action_noise(seed, np.arange(M * B * T), 2, stream=0).reshape(M, B, T, 2)
    spec = op.PolicySpec(2, 2, hidden)
    theta = spec.from_ordered_dict(policy.get_param_values())
    ref = pr.rollout(spec, np.tile(theta, (M, 1)), sampler.goals, start, noise, reward_type='dense')
    cat = lambda key: np.concatenate([p[key] for i in range(M) for p in paths[i]])
    np.testing.assert_allclose(cat('actions'), ref['act'], atol=5e-6)
    np.testing.assert_allclose(cat('rewards'), ref['rew'], atol=5e-6)


def test_trainer_with_device_rollouts(emu):
    run_trainer_scenario(n_itr=1, device_rollouts=True)


def test_meta_sample_processor_api(emu):
    run_processor_scenario()


def test_policy_algo_api(emu):
    run_algo_scenario()


def test_trainer_end_to_end(emu):
    run_trainer_scenario(n_itr=1)


def run_get_actions_scenario(M=3, B=7, O=5, A=3, hidden=(32, 32)):
    """MetaGaussianMLPPolicy.get_actions: device means == oracle forward; agent_infos layout of the reference
    (reference tests/test_policies.py:43-64: dist info from get_actions is consistent with the parameters)."""
    from oracle import policy as op
    from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
    np.random.seed(11)
    policy = MetaGaussianMLPPolicy(name='p', obs_dim=O, action_dim=A, meta_batch_size=M, hidden_sizes=hidden)
    spec = op.PolicySpec(O, A, hidden)
    theta = spec.from_ordered_dict(policy.get_param_values())
    obs = [np.random.randn(B, O).astype(np.float32) for _ in range(M)]
    actions, infos = policy.get_actions(obs)
    assert len(actions) == M and actions[0].shape == (B, A) and len(infos) == M and len(infos[0]) == B
    for i in range(M):
        mu, s, _ = op.forward(spec, theta, obs[i], clip_log_std=True)
        np.testing.assert_allclose(np.stack([d['mean'] for d in infos[i]]), mu, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(infos[i][0]['log_std'], s, rtol=1e-6)
    # post-update: per-task parameters
    new = []
    for i in range(M):
        d = policy.get_param_values()
        for k in d:
            d[k] = d[k] + 0.01 * (i + 1)
        new.append(d)
    policy.update_task_parameters(new)
    actions, infos = policy.get_actions(obs)
    for i in range(M):
        mu, s, _ = op.forward(spec, spec.from_ordered_dict(new[i]), obs[i], clip_log_std=False)
        np.testing.assert_allclose(np.stack([d['mean'] for d in infos[i]]), mu, rtol=1e-5, atol=1e-6)
    a, info = policy.get_action(obs[0][0], task=1)
    assert a.shape == (A,) and set(info.keys()) == {'mean', 'log_std'}


def test_get_actions(emu):
    run_get_actions_scenario()
    run_get_actions_scenario(M=2, B=5, O=9, A=10, hidden=(16, 24, 16))     # the layer-by-layer kernels' forward


def run_baseline_fit_predict_scenario():
    """Baseline.fit / predict / get_param_values / set_params used standalone (reference tests/test_baselines.py:67-98)."""
    from promp_amd.baselines.linear_baseline import LinearFeatureBaseline, LinearTimeBaseline
    from promp_amd.baselines.zero_baseline import ZeroBaseline
    rng = np.random.RandomState(2)
    paths = []
    for _ in range(6):
        T = rng.randint(20, 40)
        obs = rng.randn(T, 3).astype(np.float32)
        rew = (rng.randn(T) + 2).astype(np.float32)
        paths.append(dict(observations=obs, actions=rng.randn(T, 1), rewards=rew, returns=sp.discount_cumsum(rew, 0.99)))
    for cls, kind in ((LinearFeatureBaseline, sp.BASELINE_LINEAR_FEATURE), (LinearTimeBaseline, sp.BASELINE_LINEAR_TIME)):
        b = cls()
        assert np.all(b.predict(paths[0]) == 0)                       # unfit => zeros (linear_baseline.py:31-32)
        b.fit(paths, target_key='returns')
        c_ref, _, _ = sp.fit_linear_baseline([p['observations'] for p in paths], [p['returns'] for p in paths], kind)
        np.testing.assert_allclose(b.get_param_values(), c_ref, rtol=1e-6, atol=1e-8)
        pred = b.predict(paths[1])
        np.testing.assert_allclose(pred, sp.predict_linear_baseline(paths[1]['observations'], c_ref, kind), rtol=1e-6, atol=1e-7)
        # fit lowers the squared error (reference tests/test_baselines.py:67-80)
        assert np.sum((pred - paths[1]['returns']) ** 2) < np.sum(paths[1]['returns'] ** 2)
        b2 = cls()
        b2.set_params(b.get_param_values())                           # round trip (tests/test_baselines.py:82-98)
        np.testing.assert_allclose(b2.predict(paths[2]), b.predict(paths[2]), rtol=0, atol=0)
    assert np.all(ZeroBaseline().predict(paths[0]) == 0)


def test_baseline_fit_predict(emu):
    run_baseline_fit_predict_scenario()


def test_host_pool_reuses_size_classes_and_caps_what_it_keeps():
    """requests of slightly different sizes (early-terminating environments: a different row count every
    iteration) must land on the same recycled buffers, not leave one page-locked buffer per distinct size behind"""
    from promp_amd._lib import _HostPool
    pool = _HostPool(max_bytes=64 << 20)
    rng = np.random.RandomState(0)
    for _ in range(100):
        n = int(rng.randint(700000, 900000))
        bufs = [pool.get((n,), np.float32) for _ in range(8)]
        assert len(set(b.ctypes.data for b in bufs)) == 8 and all(b.shape == (n,) for b in bufs)
        del bufs
    assert pool.retained_owners() <= 16 and pool.retained_bytes() <= 64 << 20
    # live views are never recycled, whatever the cap; free owners beyond the cap are dropped
    held = [pool.get((n,), np.float32) for n in range(100000, 3000000, 100000)]
    for i, b in enumerate(held):
        b[:] = i
    assert len(set(b.ctypes.data for b in held)) == len(held) and all(float(b[0]) == i for i, b in enumerate(held))
    del held, b
    pool.get((10,), np.float32)
    assert pool.retained_bytes() <= 64 << 20


def test_host_pool_evicts_the_least_recently_used_free_buffers_first(monkeypatch):
    """over the cap the free owners go in the order they were last handed out, whatever their size class (the
    eviction used to walk the classes largest first); owners somebody references never go; PROMP_HOST_POOL_MB sets the cap"""
    from promp_amd._lib import _HostPool
    pool = _HostPool(max_bytes=4 << 20)
    old_big = pool.get((1 << 19,), np.float32)          # 2 MB class, handed out first
    small = pool.get((1 << 18,), np.float32)            # 1 MB class
    big_class = _HostPool.size_class(4 << 19)
    del old_big, small                                  # both free; the big one is the older
    again = pool.get((1 << 18,), np.float32)            # the 1 MB owner again: now the most recent
    other = pool.get((300000,), np.float32)             # a third owner: over the cap -> the oldest free owner (the big one) goes
    assert all(k[0] != big_class or not v for k, v in pool._owners.items()) and pool.retained_bytes() <= 4 << 20
    held = [pool.get((1 << 19,), np.float32) for _ in range(3)]     # referenced owners stay, cap or not
    assert pool.retained_owners() == 5 and pool.retained_bytes() > 4 << 20
    del held, again, other
    monkeypatch.setenv('PROMP_HOST_POOL_MB', '2')
    assert _HostPool()._max_bytes == 2 << 20


def test_upload_sources_stay_referenced_until_a_call_has_waited_for_the_device(emu):
    """promp_upload_step returns with its copies enqueued; their page-locked sources may be recycled only after a
    call that waited for the device -- a later upload proves nothing.  Three uploads of one step with no such call in between make
    the third wait itself; a download in between releases the older sources."""
    from promp_amd import synthetic
    rng = np.random.RandomState(5)
    M, P, T, O, A, hidden = 2, 2, 8, 4, 2, (8, 8)
    theta = synthetic.init_theta(rng, O, hidden, A)
    ctx = _lib.Context(M, O, A, hidden, 1, max_rows=M * P * T, max_paths=M * P)
    fl = _lib.flatten_paths(synthetic.make_paths(rng, theta, M, P, T, O, A, hidden))
    up = lambda: ctx.upload_step(0, fl['task_path_offsets'], fl['path_row_offsets'], fl['obs'], fl['rew'], fl['act'], fl['old_mean'], fl['old_log_std'])
    calls = []
    real = ctx._call
    ctx._call = lambda name, *a: (calls.append(name), real(name, *a))[1]
    up(); up()
    assert len(ctx._upload_refs[0]) == 2 and 'promp_sync' not in calls
    up()                                                # the third in a row: waits, then holds only its own sources
    assert calls.count('promp_sync') == 1 and len(ctx._upload_refs[0]) == 1
    ctx.process_samples(0)
    ctx.download_processed(0)                           # waited for the device: what was enqueued before is done
    up()
    assert len(ctx._upload_refs[0]) == 1 and calls.count('promp_sync') == 1
    ctx.close()


def test_hidden_nonlinearity_argument_is_honoured_or_refused(emu):
    """policies/base.py:31 / networks/mlp.py:47: hidden_nonlinearity is tf.tanh by default, any TF function or None (linear hidden
    layers).  The plugin class maps 'tanh' / 'relu' / None (and callables of those names) onto the device kernels and refuses the
    rest by name -- in particular None is NOT run as tanh (Missing #4)."""
    from oracle import policy as op
    from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
    obs = [np.random.RandomState(3).randn(5, 4).astype(np.float32) for _ in range(2)]

    def relu(x):
        return x
    for arg, kind in ((None, 'identity'), ('relu', 'relu'), (relu, 'relu'), ('tanh', 'tanh')):
        np.random.seed(4)
        pol = MetaGaussianMLPPolicy(name='p', obs_dim=4, action_dim=2, meta_batch_size=2, hidden_sizes=(16, 16), hidden_nonlinearity=arg)
        assert pol.hidden_nonlinearity == kind
        pol.switch_to_pre_update()
        _, infos = pol.get_actions(obs)
        spec = op.PolicySpec(4, 2, (16, 16), hidden_act=kind)
        theta = spec.from_ordered_dict(pol.get_param_values())
        for i in range(2):
            mean, _, _ = op.forward(spec, theta, obs[i], False)
            np.testing.assert_allclose(np.stack([inf['mean'] for inf in infos[i]]), mean, atol=2e-6)
        pol.session._drop()
    with pytest.raises(_lib.PrompError, match='unsupported'):
        MetaGaussianMLPPolicy(name='p', obs_dim=4, action_dim=2, meta_batch_size=2, hidden_sizes=(16, 16), hidden_nonlinearity='elu')


def test_host_paths_give_process_samples_the_same_results_as_plain_dicts(emu):
    """path dicts that are views of the flat page-locked arrays (what MetaSampler.obtain_samples returns for a
    fixed-horizon environment) against the same batch as independent arrays -- every field of every task's samples data, the
    per-path side effect and the baseline coefficients are equal, and the fast route did not concatenate anything"""
    from collections import OrderedDict
    from promp_amd import synthetic
    from promp_amd.baselines.linear_baseline import LinearFeatureBaseline
    from promp_amd.samplers.meta_sample_processor import MetaSampleProcessor
    from promp_amd.samplers.meta_sampler import HostPaths, slab_backed
    rng = np.random.RandomState(7)
    M, P, T, O, A = 3, 4, 12, 5, 2
    theta = synthetic.init_theta(rng, O, (8, 8), A)
    plain = synthetic.make_paths(rng, theta, M, P, T, O, A, (8, 8))
    copy = OrderedDict((i, [dict(observations=p['observations'].copy(), actions=p['actions'].copy(), rewards=p['rewards'].copy(),
                                 env_infos={}, agent_infos={k: v.copy() for k, v in p['agent_infos'].items()}) for p in pl])
                       for i, pl in plain.items())
    fast = slab_backed(copy)
    assert isinstance(fast, HostPaths) and fast.flat_if_intact() is fast.flat
    kw = dict(discount=0.97, gae_lambda=0.9, normalize_adv=True)
    pa, pb = MetaSampleProcessor(baseline=LinearFeatureBaseline(), **kw), MetaSampleProcessor(baseline=LinearFeatureBaseline(), **kw)
    out_plain, out_fast = pa.process_samples(plain), pb.process_samples(fast)
    for a, b in zip(out_plain, out_fast):
        assert set(a.keys()) == set(b.keys())
        for key in ('observations', 'actions', 'rewards', 'returns', 'advantages', 'adj_avg_rewards'):
            np.testing.assert_array_equal(np.asarray(a[key]), np.asarray(b[key]), err_msg=key)
        for key in ('mean', 'log_std'):
            np.testing.assert_array_equal(a['agent_infos'][key], b['agent_infos'][key])
    for i in range(M):
        for p, q in zip(plain[i], fast[i]):
            np.testing.assert_array_equal(p['returns'], q['returns'])
            np.testing.assert_array_equal(p['advantages'], q['advantages'])
    np.testing.assert_array_equal(pa.baseline.get_param_values(), pb.baseline.get_param_values())
    assert np.shares_memory(out_fast[0]['observations'], fast.flat['obs'])       # the samples data are views of the same arrays
    # The flat arrays go to the device BEFORE the identities of the path dicts are checked (the check runs under the copies): a batch
    # somebody edited since the sampler built it -- a replaced rewards array, a dropped path -- must still be processed as the dicts
    # stand, not as the stale flat arrays say.
    for edit in ('replace', 'drop'):
        rng2 = np.random.RandomState(8)
        plain2 = synthetic.make_paths(rng2, theta, M, P, T, O, A, (8, 8))
        fast2 = slab_backed(OrderedDict((i, [dict(p, agent_infos=dict(p['agent_infos'])) for p in pl]) for i, pl in plain2.items()))
        assert fast2.flat_if_intact() is fast2.flat
        for batch in (plain2, fast2):
            if edit == 'replace':
                batch[1][2]['rewards'] = batch[1][2]['rewards'] * 3.0 + 1.0
            else:
                batch[2].pop()
        assert fast2.flat_if_intact() is None
        pc, pd = MetaSampleProcessor(baseline=LinearFeatureBaseline(), **kw), MetaSampleProcessor(baseline=LinearFeatureBaseline(), **kw)
        for a, b in zip(pc.process_samples(plain2), pd.process_samples(fast2)):
            for key in ('observations', 'actions', 'rewards', 'returns', 'advantages', 'adj_avg_rewards'):
                np.testing.assert_array_equal(np.asarray(a[key]), np.asarray(b[key]), err_msg='%s after %s' % (key, edit))
        np.testing.assert_array_equal(pc.baseline.get_param_values(), pd.baseline.get_param_values())
