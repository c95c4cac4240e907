"""Developer measurement for SURVEY 8f rows 1/3: env-steps/s of rollout collection on the point-mass meta-environment,
host loop (MetaSampler: one device policy query per environment step) vs device rollout (DevicePointEnvSampler)."""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from promp_amd.envs.point_env import MetaPointEnv
from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
from promp_amd.samplers.device_point_sampler import DevicePointEnvSampler
from promp_amd.samplers.meta_sampler import MetaSampler

for M, B, T in ((4, 20, 100), (40, 20, 200)):
    np.random.seed(0)
    env = MetaPointEnv()
    policy = MetaGaussianMLPPolicy(name='p', obs_dim=2, action_dim=2, meta_batch_size=M, hidden_sizes=(32, 32))
    policy.switch_to_pre_update()
    for name, cls, reps in (('host loop   ', MetaSampler, 2), ('device kernel', DevicePointEnvSampler, 20)):
        s = cls(env=env, policy=policy, rollouts_per_meta_task=B, meta_batch_size=M, max_path_length=T)
        s.update_tasks()
        s.obtain_samples()
        t0 = time.time()
        for _ in range(reps):
            s.obtain_samples()
        dt = (time.time() - t0) / reps
        print('M=%d B=%d T=%d  %s %9.3f ms per sampling step  %12.0f env-steps/s' % (M, B, T, name, 1e3 * dt, M * B * T / dt))
