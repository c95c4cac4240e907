#!/usr/bin/env python
"""ProMP on the 2-D point-mass meta-environment -- the promp_amd counterpart of the reference's
run_scripts/pro-mp_run_point_mass.py (same config keys; BASELINE.json configs[0] shapes: obs 2, act 2, 2x32 MLP).

    python run_scripts/pro-mp_run_point_mass.py [--config_file cfg.json] [--dump_path DIR] [--n_itr N]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from promp_amd.baselines.linear_baseline import LinearFeatureBaseline  # noqa: E402
from promp_amd.envs.normalized_env import normalize  # noqa: E402
from promp_amd.envs.point_env import MetaPointEnvCorner  # noqa: E402
from promp_amd.meta_algos.pro_mp import ProMP  # noqa: E402
from promp_amd.meta_trainer import Trainer  # noqa: E402
from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy  # noqa: E402
from promp_amd.samplers.meta_sample_processor import MetaSampleProcessor  # noqa: E402
from promp_amd.samplers.device_point_sampler import DevicePointEnvSampler  # noqa: E402
from promp_amd.samplers.meta_sampler import MetaSampler  # noqa: E402
from promp_amd.utils import logger  # noqa: E402

DEFAULT = {
    'seed': 1, 'baseline': 'LinearFeatureBaseline', 'env': 'MetaPointEnvCorner', 'reward_type': 'sparse', 'normalize_env': True,
    'rollouts_per_meta_task': 20, 'max_path_length': 100, 'parallel': False,
    'discount': 0.99, 'gae_lambda': 1, 'normalize_adv': True,
    'hidden_sizes': (32, 32),
    'inner_lr': 0.1, 'learning_rate': 1e-3, 'num_promp_steps': 5, 'clip_eps': 0.3, 'target_inner_step': 0.01,
    'init_inner_kl_penalty': 5e-4, 'adaptive_inner_kl_penalty': False,
    'n_itr': 100, 'meta_batch_size': 4, 'num_inner_grad_steps': 1,
    'device_rollouts': False,      # True: the environment itself runs on the GPU (samplers/device_point_sampler.py)
    'device_noise': False,         # with device_rollouts: exploration noise drawn on the device (Philox) instead of NumPy's RNG
}


def main(config):
    np.random.seed(config['seed'])
    baseline = {'LinearFeatureBaseline': LinearFeatureBaseline}[config['baseline']]()
    env = MetaPointEnvCorner(reward_type=config.get('reward_type', 'sparse'))
    if config.get('normalize_env', True):
        env = normalize(env)                      # as the reference script (pro-mp_run_point_mass.py:27-28)
    policy = MetaGaussianMLPPolicy(name='meta-policy', obs_dim=2, action_dim=2, meta_batch_size=config['meta_batch_size'],
                                   hidden_sizes=config['hidden_sizes'])
    sampler_kwargs = dict(env=env, policy=policy, rollouts_per_meta_task=config['rollouts_per_meta_task'],
                          meta_batch_size=config['meta_batch_size'], max_path_length=config['max_path_length'],
                          parallel=config['parallel'])
    if config.get('device_rollouts'):
        sampler = DevicePointEnvSampler(device_noise=bool(config.get('device_noise')), **sampler_kwargs)
    else:
        sampler = MetaSampler(**sampler_kwargs)
    sample_processor = MetaSampleProcessor(baseline=baseline, discount=config['discount'], gae_lambda=config['gae_lambda'],
                                           normalize_adv=config['normalize_adv'])
    algo = ProMP(policy=policy, inner_lr=config['inner_lr'], meta_batch_size=config['meta_batch_size'],
                 num_inner_grad_steps=config['num_inner_grad_steps'], learning_rate=config['learning_rate'],
                 num_ppo_steps=config['num_promp_steps'], clip_eps=config['clip_eps'],
                 target_inner_step=config['target_inner_step'], init_inner_kl_penalty=config['init_inner_kl_penalty'],
                 adaptive_inner_kl_penalty=config['adaptive_inner_kl_penalty'])
    trainer = Trainer(algo=algo, policy=policy, env=env, sampler=sampler, sample_processor=sample_processor,
                      n_itr=config['n_itr'], num_inner_grad_steps=config['num_inner_grad_steps'])
    trainer.train()
    return policy


if __name__ == '__main__':
    ap = argparse.ArgumentParser(description='ProMP: Proximal Meta-Policy Search (MI355X)')
    ap.add_argument('--config_file', type=str, default='')
    ap.add_argument('--dump_path', type=str, default='')
    ap.add_argument('--n_itr', type=int, default=None)
    ap.add_argument('--quiet', action='store_true')
    args = ap.parse_args()
    cfg = dict(DEFAULT)
    if args.config_file:
        cfg.update(json.load(open(args.config_file)))
    if args.n_itr is not None:
        cfg['n_itr'] = args.n_itr
    logger.configure(dir=args.dump_path or None, snapshot_mode='last_gap', snapshot_gap=50, quiet=args.quiet)
    if args.dump_path:
        json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, open(os.path.join(args.dump_path, 'params.json'), 'w'))
    main(cfg)
