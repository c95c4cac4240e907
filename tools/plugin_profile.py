"""developer tool: cProfile of the plugin-level step (device-resident batch and host path dicts) at config-3 size"""
import cProfile, pstats, io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from promp_amd import _lib, synthetic
from promp_amd.baselines.linear_baseline import LinearFeatureBaseline
from promp_amd.meta_algos.pro_mp import ProMP
from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
from promp_amd.samplers.device_point_sampler import DevicePaths
from promp_amd.samplers.meta_sample_processor import MetaSampleProcessor
from promp_amd.utils import logger as plog
plog.configure(quiet=True)
cfg = synthetic.CONFIGS[3]
M, P, T, O, A, hidden = cfg['M'], cfg['P'], cfg['T'], cfg['O'], cfg['A'], cfg['hidden']
theta0 = synthetic.init_theta(np.random.RandomState(3000), O, hidden, A)
policy = MetaGaussianMLPPolicy(name='p', obs_dim=O, action_dim=A, meta_batch_size=M, hidden_sizes=hidden, rank=0, world=1, device_id=0)
policy.set_params(policy._unflatten(theta0))
proc = MetaSampleProcessor(baseline=LinearFeatureBaseline(), discount=0.99, gae_lambda=1.0, normalize_adv=True)
algo = ProMP(policy=policy, inner_lr=0.1, meta_batch_size=M, num_inner_grad_steps=1, learning_rate=1e-3, num_ppo_steps=5, clip_eps=0.3,
             target_inner_step=0.01, init_inner_kl_penalty=5e-4, adaptive_inner_kl_penalty=True)
rng = np.random.RandomState(4321)
p0 = synthetic.make_paths(rng, theta0, M, P, T, O, A, hidden)
policy.switch_to_pre_update()
sd0 = proc.process_samples(p0, log=False)
algo._adapt(sd0)
th1 = np.stack([np.concatenate([v.reshape(-1) for v in d.values()]) for d in policy.policies_params_vals]).astype(np.float32)
p1 = synthetic.make_paths(rng, th1, M, P, T, O, A, hidden)
sess = policy.session
dev = []
for slot, paths in ((0, p0), (1, p1)):
    fl = _lib.flatten_paths(paths)
    sess.upload_flat(slot, fl)
    dp = DevicePaths(paths); dp.device_ref = (sess.serial, sess.upload_serial[slot], slot); dp.flat = fl
    dev.append(dp)

def step(b0, b1):
    policy.switch_to_pre_update()
    s0 = proc.process_samples(b0, log=False)
    algo._adapt(s0)
    s1 = proc.process_samples(b1, log=False)
    algo.optimize_policy([s0, s1], log=False)

from collections import OrderedDict
from promp_amd.samplers.meta_sampler import slab_backed
hp = [slab_backed(OrderedDict((i, [dict(p, agent_infos=dict(p['agent_infos'])) for p in pl]) for i, pl in px.items())) for px in (p0, p1)]
for name, (b0, b1) in (('host_paths (HostPaths)', hp), ('host_paths (plain dicts)', (p0, p1))):
    for _ in range(2):
        step(b0, b1)
    t0 = time.perf_counter()
    for _ in range(5):
        step(b0, b1)
    print('== %s: %.3f ms per step' % (name, 1e3 * (time.perf_counter() - t0) / 5))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        step(b0, b1)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22); print('\n'.join(s.getvalue().split('\n')[:48]))
