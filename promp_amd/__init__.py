"""promp_amd -- MI355X-native ProMP/MAML hot path behind the reference's plugin API.

    from promp_amd.baselines.linear_baseline import LinearFeatureBaseline
    from promp_amd.samplers.meta_sample_processor import MetaSampleProcessor
    from promp_amd.policies.meta_gaussian_mlp_policy import MetaGaussianMLPPolicy
    from promp_amd.meta_algos.pro_mp import ProMP
    from promp_amd.meta_trainer import Trainer

mirror meta_policy_search.* of jonasrothfuss/ProMP (same constructor arguments, same method names and return
conventions); the arithmetic runs in promp_amd/libpromp_hip.so (hand-written HIP for gfx950) through ctypes.
No PyTorch, no TensorFlow, no CPU fallback.
"""
__all__ = ['_lib', 'session', 'synthetic']
