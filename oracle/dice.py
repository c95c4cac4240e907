"""Oracle for DICE-MAML: the DiCE sample processor and the DiCE objective with its exact first and
second derivatives, float64 NumPy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates (paths relative to /root/reference/meta_policy_search/):
  * samplers/dice_sample_processor.py:96-191   discounted rewards r_t gamma^t, baseline fit on them, adjusted rewards,
                                               zero padding to max_path_length + mask, normalisation over the PADDED array
  * meta_algos/dice_maml.py:39-45, 245-258     obj = -mean_{p,t}( magic_box(tau)_{p,t} * adj_{p,t} * mask_{p,t} ),
                                               tau_{p,t} = sum_{t' <= t} log pi(a_{p,t'} | o_{p,t'}),
                                               magic_box(x) = exp(x - stop_gradient(x))
  * meta_algos/dice_maml.py:84-152             MAML graph: DiCE objective as inner AND outer objective, no KL terms

Derivatives of the magic box (value 1, derivative = derivative of its argument, second derivative = outer product + Hessian):
  value     L       = -(1/N) sum_{p,t} R_{p,t} m_{p,t}                                   N = P * max_path_length
  gradient  dL      = -(1/N) sum_{p,t'} w_{p,t'} dlogpi_{p,t'},     w_{p,t'} = sum_{t >= t'} R_{p,t} m_{p,t}
  Hessian-vector    H v = -(1/N) [ sum w d^2 logpi v  +  sum_{p,t'} u_{p,t'} dlogpi_{p,t'} ],
                    u_{p,t'} = sum_{t >= t'} R_{p,t} m_{p,t} C_{p,t},   C_{p,t} = sum_{t'' <= t} dlogpi_{p,t''} . v
i.e. the log-likelihood objective with per-row weights w (gradient and R-operator pass as in oracle/promp.py) plus a second
log-likelihood GRADIENT whose weights u couple the time steps of a path.  A flat "slab" stores only the valid rows
(mask == 1): padded rows have mask 0 and, lying behind the valid ones, never enter a valid row's tau.  Weights are stored
pre-scaled by rows / N so that the slab mean (1 / rows) reproduces the reference's mean over the padded array.
"""
import numpy as np

from . import policy as op
from . import promp as pm
from . import sample_processing as sp


# ---- sample processing --------------------------------------------------------------------------------------------------

def process_samples_dice(paths, max_path_length, baseline_kind=sp.BASELINE_LINEAR_TIME, discount=0.99, normalize_adv=True,
                         positive_adv=False, reg_coeff=1e-5, return_baseline_kind=None, gae_lambda=1.0):
    """DiceSampleProcessor._compute_samples_data for ONE task (dice_sample_processor.py:96-131).
    -> dict(mask, observations, actions, rewards, adjusted_rewards, agent_infos) padded to [P, max_path_length, ...];
    with return_baseline_kind also 'advantages': GAE advantages from a second baseline fitted on the returns, padded and
    normalised / shifted over the padded array (:125-127, 196-238)"""
    T = int(max_path_length)
    disc = np.cumprod(np.concatenate([np.ones(1), np.ones(T - 1) * discount]))                 # :150
    targets = []
    for p in paths:
        n = len(p['rewards'])
        assert n <= T
        targets.append(np.asarray(p['rewards'], dtype=np.float64) * disc[:n])                    # :154
    if baseline_kind == sp.BASELINE_ZERO:
        preds = [np.zeros(len(t)) for t in targets]
    else:
        obs = [np.asarray(p['observations']) for p in paths]      # (dtype kept: the reference squares float32 observations in float32)
        w, _, _ = sp.fit_linear_baseline(obs, targets, baseline_kind, reg_coeff)                # baseline.fit(target_key=...)
        preds = [sp.predict_linear_baseline(o, w, baseline_kind) for o in obs]
    adj = [t - b for t, b in zip(targets, preds)]                                               # :163

    def pad(a):
        a = np.asarray(a)
        width = ((0, T - a.shape[0]),) + ((0, 0),) * (a.ndim - 1)
        return np.pad(a, width, mode='constant')
    out = dict(mask=np.stack([pad(np.ones(len(t))) for t in targets]),
               observations=np.stack([pad(p['observations']) for p in paths]),
               actions=np.stack([pad(p['actions']) for p in paths]),
               rewards=np.stack([pad(p['rewards']) for p in paths]),
               adjusted_rewards=np.stack([pad(a) for a in adj]),
               agent_infos=dict(mean=np.stack([pad(p['agent_infos']['mean']) for p in paths]),
                                log_std=np.stack([pad(p['agent_infos']['log_std']) for p in paths])))
    a = out['adjusted_rewards']
    if normalize_adv:                                                                           # utils/utils.py:59-66 on the padded array
        a = (a - a.mean()) / (a.std() + 1e-8)
    if positive_adv:
        a = (a - a.min()) + 1e-8
    out['adjusted_rewards'] = a
    if return_baseline_kind is not None:
        rews = [np.asarray(p['rewards'], dtype=np.float64) for p in paths]
        rets = [sp.discount_cumsum(r, discount) for r in rews]                                   # :206
        if return_baseline_kind == sp.BASELINE_ZERO:
            base = [np.zeros(len(r)) for r in rews]
        else:
            obs = [np.asarray(p['observations']) for p in paths]
            w, _, _ = sp.fit_linear_baseline(obs, rets, return_baseline_kind, reg_coeff)          # :209-210
            base = [sp.predict_linear_baseline(o, w, return_baseline_kind) for o in obs]
        adv = np.stack([pad(sp.compute_advantages(r, b, discount, gae_lambda)) for r, b in zip(rews, base)])   # :213-227
        if normalize_adv:
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        if positive_adv:
            adv = (adv - adv.min()) + 1e-8
        out['advantages'] = adv
    return out


def to_slab(sd):
    """Padded DiCE samples of one task -> flat slab of the valid rows with the DiCE weights.
    'dice_rw' = adjusted reward * rows / N per valid row, 'advantages' = its suffix sum within the path (= w * rows / N)."""
    mask = np.asarray(sd['mask']) > 0.5
    P, T = mask.shape
    lens = mask.sum(axis=1).astype(int)
    assert all(mask[p, :lens[p]].all() for p in range(P)), 'padding must follow the valid steps'
    rows = int(lens.sum())
    scale = rows / float(P * T)
    sel = lambda x: np.concatenate([np.asarray(x)[p, :lens[p]] for p in range(P)])
    rw = sel(sd['adjusted_rewards']).astype(np.float64) * scale
    off = np.concatenate([[0], np.cumsum(lens)]).astype(int)
    w = np.concatenate([np.cumsum(rw[off[p]:off[p + 1]][::-1])[::-1] for p in range(P)])
    out = dict(observations=sel(sd['observations']), actions=sel(sd['actions']), advantages=w, dice_rw=rw,
               path_row_offsets=off,
               agent_infos=dict(mean=sel(sd['agent_infos']['mean']), log_std=sel(sd['agent_infos']['log_std'])))
    if 'advantages' in sd:      # VPG-DiCE: the outer objective's weights; the mean over the padded array becomes the slab mean
        out['vpg_advantages'] = sel(sd['advantages']).astype(np.float64) * scale
    return out


# ---- objective ------------------------------------------------------------------------------------------------------------

def loss_value(slab):
    """-(1/N) sum R m  (the magic box evaluates to one)"""
    return -float(np.sum(slab['dice_rw'])) / len(slab['dice_rw'])


def loss_and_grad(spec, theta, slab, clip_log_std):
    r = pm.loss_and_grad(spec, theta, slab, 'loglik', clip_log_std)
    return dict(loss=loss_value(slab), grad=r['grad'])


def row_tangent(spec, theta, slab, v, clip_log_std):
    """dlogpi_row . v for every row (forward-mode through forward() and the log-likelihood)"""
    obs = np.asarray(slab['observations'], dtype=np.float64)
    act = np.asarray(slab['actions'], dtype=np.float64)
    mu, s, cache = op.forward(spec, theta, obs, clip_log_std)
    acts, parts = cache['acts'], cache['parts']
    vparts = spec.unflatten(np.asarray(v, dtype=np.float64))
    nl = len(spec.layer_shapes)
    Rx = np.zeros_like(acts[0])
    for li in range(nl):
        Rz = Rx @ parts[2 * li] + acts[li] @ vparts[2 * li] + vparts[2 * li + 1]
        Rx = (1.0 - acts[li + 1] ** 2) * Rz if li < nl - 1 else Rz
    Rs = vparts[-1].reshape(-1) * cache['s_mask']
    e = np.exp(-s)
    z = (act - mu) * e
    return np.sum(z * e * Rx + (z ** 2 - 1.0) * Rs, axis=1)


def coupling_weights(slab, c):
    """u_{t'} = sum_{t >= t'} rw_t C_t,  C_t = sum_{t'' <= t} c_t''   within each path"""
    off, rw = slab['path_row_offsets'], slab['dice_rw']
    u = np.zeros_like(rw)
    for p in range(len(off) - 1):
        a, b = off[p], off[p + 1]
        C = np.cumsum(c[a:b])
        u[a:b] = np.cumsum((rw[a:b] * C)[::-1])[::-1]
    return u


def hvp(spec, theta, slab, v, clip_log_std):
    """(d^2 L_dice / d theta^2) v"""
    h1 = pm.hvp(spec, theta, slab, v, 'loglik', clip_log_std)
    u = coupling_weights(slab, row_tangent(spec, theta, slab, v, clip_log_std))
    h2 = pm.loss_and_grad(spec, theta, dict(slab, advantages=u), 'loglik', clip_log_std)['grad']
    return h1 + h2


def adapt(spec, thetas_tasks, slabs, step_sizes):
    """DICEMAML inner step (dice_maml.py:47-82: explicit per-task parameters => raw log_std)"""
    return [np.asarray(th, dtype=np.float64) - step_sizes * loss_and_grad(spec, th, sl, False)['grad']
            for th, sl in zip(thetas_tasks, slabs)]


def meta_objective_and_grad(spec, theta, all_slabs, step_sizes, want_grad=True, outer='dice'):
    """DICEMAML.build_graph (dice_maml.py:84-152) at theta: mean over tasks of the DiCE objective of the last step's samples
    at the adapted parameters, with its exact gradient through the K adaptation steps.
    outer='vpg': VPG_DICEMAML.build_graph (vpg_dice_maml.py:35-113) -- the same DiCE inner steps under the outer objective
    -mean(log pi * advantage * mask) of the last step's samples."""
    K, M = len(all_slabs) - 1, len(all_slabs[0])
    theta = np.asarray(theta, dtype=np.float64)
    loss, grad, adapted = 0.0, np.zeros_like(theta), []
    for i in range(M):
        thetas = [theta]
        for k in range(K):
            thetas.append(thetas[k] - step_sizes * loss_and_grad(spec, thetas[k], all_slabs[k][i], k == 0)['grad'])
        if outer == 'vpg':
            last = all_slabs[K][i]
            r = pm.loss_and_grad(spec, thetas[K], dict(last, advantages=last['vpg_advantages']), 'loglik', False)
        else:
            r = loss_and_grad(spec, thetas[K], all_slabs[K][i], False)
        loss += r['loss']
        adapted.append(thetas[K])
        if want_grad:
            lam = r['grad']
            for k in range(K - 1, -1, -1):
                lam = lam - hvp(spec, thetas[k], all_slabs[k][i], step_sizes * lam, k == 0)
            grad += lam
    out = dict(loss=loss / M, adapted=adapted)
    if want_grad:
        out['grad'] = grad / M
    return out
