#!/usr/bin/env python
"""TRPO-MAML / E-MAML on the 2-D point-mass meta-environment -- the promp_amd counterpart of the reference's
run_scripts/maml_run_mujoco.py and e-maml_run_mujoco.py (same config keys; the two differ in `exploration` only,
maml_run_mujoco.py:60 / e-maml_run_mujoco.py:60) on the one environment of the reference that needs no MuJoCo (BASELINE.json
configs[0] shapes: obs 2, act 2, 2x32 MLP; configs[4] is this algorithm at HalfCheetah's shapes).

    python run_scripts/maml_run_point_mass.py [--config_file cfg.json] [--dump_path DIR] [--n_itr N] [--exploration] [--hvp exact]
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
from promp_amd.meta_algos.trpo_maml import TRPOMAML  # noqa: E402
from promp_amd.meta_trainer import Trainer  # noqa: E402
from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy  # noqa: E402
from promp_amd.samplers.meta_sample_processor import MetaSampleProcessor  # noqa: E402
from promp_amd.samplers.meta_sampler import MetaSampler  # noqa: E402
from promp_amd.utils import logger  # noqa: E402
from promp_amd.utils.utils import ClassEncoder, set_seed  # noqa: E402

DEFAULT = {
    'seed': 1, 'baseline': 'LinearFeatureBaseline', 'env': 'MetaPointEnvCorner', 'reward_type': 'dense',
    # sampler / sample processor (maml_run_mujoco.py:101-110)
    'rollouts_per_meta_task': 20, 'max_path_length': 100, 'parallel': False,
    'discount': 0.99, 'gae_lambda': 1, 'normalize_adv': True,
    # policy
    'hidden_sizes': (32, 32), 'learn_std': True,
    # TRPO-MAML (maml_run_mujoco.py:116-123)
    'inner_lr': 0.1, 'step_size': 0.01, 'n_itr': 100, 'meta_batch_size': 4, 'num_inner_grad_steps': 1,
    'inner_type': 'log_likelihood',
    'exploration': False,          # True: E-MAML (e-maml_run_mujoco.py:60)
    'hvp_approach': 'finite_difference',   # the reference's; 'exact': the device's exact constraint product
}


def main(config):
    set_seed(config['seed'])
    baseline = {'LinearFeatureBaseline': LinearFeatureBaseline}[config['baseline']]()
    env = normalize({'MetaPointEnvCorner': MetaPointEnvCorner}[config['env']](reward_type=config.get('reward_type', 'dense')))
    policy = MetaGaussianMLPPolicy(name='meta-policy', obs_dim=int(np.prod(env.observation_space.shape)),
                                   action_dim=int(np.prod(env.action_space.shape)), meta_batch_size=config['meta_batch_size'],
                                   hidden_sizes=config['hidden_sizes'], learn_std=config.get('learn_std', True))
    sampler = MetaSampler(env=env, policy=policy, rollouts_per_meta_task=config['rollouts_per_meta_task'],
                          meta_batch_size=config['meta_batch_size'], max_path_length=config['max_path_length'],
                          parallel=config['parallel'])
    sample_processor = MetaSampleProcessor(baseline=baseline, discount=config['discount'], gae_lambda=config['gae_lambda'],
                                           normalize_adv=config['normalize_adv'])
    algo = TRPOMAML(policy=policy, step_size=config['step_size'], inner_type=config['inner_type'], inner_lr=config['inner_lr'],
                    meta_batch_size=config['meta_batch_size'], num_inner_grad_steps=config['num_inner_grad_steps'],
                    exploration=bool(config.get('exploration', False)), hvp_approach=config.get('hvp_approach', 'finite_difference'))
    trainer = Trainer(algo=algo, policy=policy, env=env, sampler=sampler, sample_processor=sample_processor,
                      n_itr=config['n_itr'], num_inner_grad_steps=config['num_inner_grad_steps'])
    trainer.train()
    return policy


if __name__ == '__main__':
    ap = argparse.ArgumentParser(description='TRPO-MAML / E-MAML (MI355X)')
    ap.add_argument('--config_file', type=str, default='', help='json file with run specifications')
    ap.add_argument('--dump_path', type=str, default='')
    ap.add_argument('--n_itr', type=int, default=None)
    ap.add_argument('--exploration', action='store_true', help='E-MAML (e-maml_run_mujoco.py)')
    ap.add_argument('--hvp', type=str, default=None, choices=['finite_difference', 'exact'])
    ap.add_argument('--quiet', action='store_true')
    args = ap.parse_args()
    cfg = dict(DEFAULT)
    if args.config_file:
        cfg.update(json.load(open(args.config_file)))
    if args.n_itr is not None:
        cfg['n_itr'] = args.n_itr
    if args.exploration:
        cfg['exploration'] = True
    if args.hvp:
        cfg['hvp_approach'] = args.hvp
    logger.configure(dir=args.dump_path or None, snapshot_mode='last_gap', snapshot_gap=50, quiet=args.quiet)
    if args.dump_path:
        json.dump(cfg, open(os.path.join(args.dump_path, 'params.json'), 'w'), cls=ClassEncoder)
    main(cfg)
