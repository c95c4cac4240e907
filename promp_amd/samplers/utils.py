"""rollout(): one episode of one environment under one policy, for evaluation scripts and notebooks (reference:
meta_policy_search/samplers/utils.py:5-71).  Not on the training path -- MetaSampler and the device samplers collect the
meta-batch -- and without the reference's viewer branch (animated / save_video drive a MuJoCo window; asked for, they raise)."""
import numpy as np


def _flat(space, x):
    """gym spaces flatten their samples (Box: ravel); the spaces of this package carry only shapes"""
    return space.flatten(x) if hasattr(space, 'flatten') else np.asarray(x).reshape(-1)


def rollout(env, agent, max_path_length=np.inf, animated=False, speedup=1, save_video=False, video_filename='sim_out.mp4',
            ignore_done=False):
    """-> dict(observations, actions, rewards, agent_infos, env_infos): per-step LISTS, as the reference returns them (flattened
    observations / actions, the agent's and the environment's info dicts; `actons`, the key the reference misspells `actions` with
    at :63, aliases the same list).  The episode ends after max_path_length steps or at the first done (unless ignore_done).
    agent.get_action(obs) -> (action, agent_info) as MetaGaussianMLPPolicy.get_action; agent.reset() is called if it exists."""
    if animated or save_video:
        raise NotImplementedError('rollout: rendering is not part of this package (the reference drives a MuJoCo viewer here)')
    observations, actions, rewards, agent_infos, env_infos = [], [], [], [], []
    o = env.reset()
    if hasattr(agent, 'reset'):
        agent.reset()
    steps = 0
    while steps < max_path_length:
        a, agent_info = agent.get_action(o)
        next_o, r, done, env_info = env.step(a)
        observations.append(_flat(env.observation_space, o))
        actions.append(_flat(env.action_space, a))
        rewards.append(r)
        agent_infos.append(agent_info)
        env_infos.append(env_info)
        steps += 1
        if done and not ignore_done:
            break
        o = next_o
    return dict(observations=observations, actions=actions, actons=actions, rewards=rewards, agent_infos=agent_infos,
                env_infos=env_infos)
