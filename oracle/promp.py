"""Oracle for reference rows a10-a13: MAML inner step, ProMP meta-objective, its exact
(second-order) gradient, Adam, KL-coefficient rule.  float64 NumPy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference builds these as a TensorFlow-1 graph and differentiates it with
``tf.gradients`` / ``AdamOptimizer.minimize``; TensorFlow is absent here, so this is a
restatement of the graph's arithmetic with hand-derived reverse mode and an R-operator
Hessian-vector product.  It is cross-checked by finite differences and torch.autograd
(tests/test_oracle_policy.py).  Restated lines (relative to
/root/reference/meta_policy_search/):

  * meta_algos/pro_mp.py:59-65          inner surrogate  -mean(ratio * adv)
  * meta_algos/base.py:192-215          theta' = theta - step_size * grad
  * meta_algos/base.py:217-242          _adapt (explicit per-task params, raw log_std)
  * meta_algos/pro_mp.py:88-155         meta-objective (clipped surrogate + inner-KL penalty)
  * meta_algos/trpo_maml.py:58-62,135   log-likelihood inner / unclipped outer variants
  * optimizers/maml_first_order_optimizer.py:82-115,146-163  E Adam epochs, loss-before, stats
  * meta_algos/pro_mp.py:201-214        KL coefficient adaptation
  * TF1 semantics: AdamOptimizer (b1=.9,b2=.999,eps=1e-8, bias-corrected lr);
    tf.minimum routes the gradient to its first argument when x <= y; tf.clip_by_value
    passes gradient inside [lo, hi]; tf.maximum(x, c) passes gradient when x >= c.
"""
import numpy as np
from .policy import forward, act_d, act_dd_over_d, LOG_2PI

INNER_RATIO = 'ratio'      # pro_mp.py:59-65
INNER_LOGLIK = 'loglik'    # trpo_maml.py:58-62
OUTER_CLIP = 'clip'        # pro_mp.py:141-145
OUTER_RATIO = 'ratio'      # trpo_maml.py:135
OUTER_KL = 'kl'            # mean KL(old || new): the TRPO constraint, trpo_maml.py:133,147,158


def _slab(slab):
    obs = np.asarray(slab['observations'], dtype=np.float64)
    act = np.asarray(slab['actions'], dtype=np.float64)
    adv = np.asarray(slab['advantages'], dtype=np.float64)
    om = np.asarray(slab['agent_infos']['mean'], dtype=np.float64)
    ols = np.broadcast_to(np.asarray(slab['agent_infos']['log_std'], dtype=np.float64), om.shape)
    return obs, act, adv, om, ols


def _backprop(spec, cache, dmu, ds):
    """Reverse mode through forward(): cotangents (dmu [N,A], ds [A]) -> flat gradient."""
    acts, parts = cache['acts'], cache['parts']
    nl = len(spec.layer_shapes)
    grads = [None] * (2 * nl + 1)
    dz = dmu * act_d(spec.output_act, acts[nl])          # output_nonlinearity (mlp.py:114-117; identity: ones)
    for li in range(nl - 1, -1, -1):
        grads[2 * li] = acts[li].T @ dz
        grads[2 * li + 1] = dz.sum(axis=0)
        if li > 0:
            dz = (dz @ parts[2 * li].T) * act_d(spec.hidden_act, acts[li])
    grads[2 * nl] = (ds * cache['s_mask']).reshape(1, -1)
    return spec.flatten(grads)


def loss_and_grad(spec, theta, slab, kind, clip_log_std, clip_eps=None, want_grad=True):
    """Per-task objective on one slab, mean KL(old||new), and their gradients wrt theta.

    kind: 'ratio' -mean(rho*A) | 'clip' -mean(min(rho*A, clip(rho)*A)) | 'loglik' -mean(logpi*A) | 'kl' mean KL
    Returns dict(loss, kl, grad, grad_kl).
    """
    obs, act, adv, om, ols = _slab(slab)
    N, A = act.shape
    mu, s, cache = forward(spec, theta, obs, clip_log_std)
    e = np.exp(-s)
    z = (act - mu) * e
    lp = -np.sum(s) - 0.5 * np.sum(z ** 2, axis=1) - 0.5 * A * LOG_2PI
    zo = (act - om) * np.exp(-ols)
    lp_old = -np.sum(ols, axis=1) - 0.5 * np.sum(zo ** 2, axis=1) - 0.5 * A * LOG_2PI
    rho = np.exp(lp - lp_old)
    # dloss/dlogpi per row ("c")
    if kind == 'ratio':
        loss = -np.mean(rho * adv)
        c = -adv * rho / N
    elif kind == 'clip':
        x = rho * adv
        y = np.clip(rho, 1.0 - clip_eps, 1.0 + clip_eps) * adv
        loss = -np.mean(np.minimum(x, y))
        c = np.where(x <= y, -adv * rho / N, 0.0)           # see module docstring (TF sub-gradients)
    elif kind == 'loglik':
        loss = -np.mean(lp * adv)
        c = -adv / N
    elif kind == 'kl':
        loss = None            # filled in below
        c = np.zeros(N)
    else:
        raise ValueError(kind)
    # KL(old || new), diagonal_gaussian.py:16-45
    so2, sn2 = np.exp(2 * ols), np.exp(2 * s)
    num = (om - mu) ** 2 + so2 - sn2
    den = 2 * sn2 + 1e-8
    kl_rows = np.sum(num / den + s - ols, axis=1)
    if kind == 'kl':
        loss = np.mean(kl_rows)
    out = dict(loss=float(loss), kl=float(np.mean(kl_rows)))
    if want_grad:
        dkl_mu = (-2.0 * (om - mu) / den) / N
        dkl_s = np.sum((-2 * sn2 * den - num * 4 * sn2) / den ** 2 + 1.0, axis=0) / N
        out['grad_kl'] = _backprop(spec, cache, dkl_mu, dkl_s)
        if kind == 'kl':
            out['grad'] = out['grad_kl']
        else:
            dmu = c[:, None] * z * e
            ds = np.sum(c[:, None] * (z ** 2 - 1.0), axis=0)
            out['grad'] = _backprop(spec, cache, dmu, ds)
    return out


def hvp(spec, theta, slab, v, kind, clip_log_std):
    """Hessian-vector product  (d^2 L / d theta^2) v  of the inner objective ('ratio'|'loglik').

    Pearlmutter R-operator: forward tangents, then the directional derivative of every
    quantity of the reverse pass.  This is the term tf.gradients produces when it
    differentiates through meta_algos/base.py:206 (second-order MAML).
    """
    obs, act, adv, om, ols = _slab(slab)
    N, A = act.shape
    mu, s, cache = forward(spec, theta, obs, clip_log_std)
    acts, parts, s_mask = cache['acts'], cache['parts'], cache['s_mask']
    vparts = spec.unflatten(np.asarray(v, dtype=np.float64))
    nl = len(spec.layer_shapes)
    # ---- R-forward
    Racts = [np.zeros_like(acts[0])]
    Rx = Racts[0]
    for li in range(nl):
        Rz = Rx @ parts[2 * li] + acts[li] @ vparts[2 * li] + vparts[2 * li + 1]
        Rx = act_d(spec.hidden_act if li < nl - 1 else spec.output_act, acts[li + 1]) * Rz
        Racts.append(Rx)
    Rmu = Racts[-1]
    Rs = vparts[-1].reshape(-1) * s_mask
    # ---- loss level
    e = np.exp(-s)
    z = (act - mu) * e
    lp = -np.sum(s) - 0.5 * np.sum(z ** 2, axis=1) - 0.5 * A * LOG_2PI
    if kind == 'ratio':
        zo = (act - om) * np.exp(-ols)
        lp_old = -np.sum(ols, axis=1) - 0.5 * np.sum(zo ** 2, axis=1) - 0.5 * A * LOG_2PI
        c = -adv * np.exp(lp - lp_old) / N
        Rlp = np.sum(z * e * Rmu + (z ** 2 - 1.0) * Rs, axis=1)
        Rc = c * Rlp
    elif kind == 'loglik':
        c = -adv / N
        Rc = np.zeros(N)
    else:
        raise ValueError(kind)
    Rzn = -Rmu * e - z * Rs
    dmu = c[:, None] * z * e
    Rdmu = Rc[:, None] * z * e + c[:, None] * (Rzn * e - z * e * Rs)
    Rds = np.sum(Rc[:, None] * (z ** 2 - 1.0) + c[:, None] * 2.0 * z * Rzn, axis=0)
    # ---- R-backward
    out = [None] * (2 * nl + 1)
    d_out = act_d(spec.output_act, mu)                   # through the output nonlinearity (identity: ones, zero curvature)
    dz, Rdz = dmu * d_out, Rdmu * d_out + dmu * act_dd_over_d(spec.output_act, mu) * Rmu
    for li in range(nl - 1, -1, -1):
        out[2 * li] = Racts[li].T @ dz + acts[li].T @ Rdz
        out[2 * li + 1] = Rdz.sum(axis=0)
        if li > 0:
            dx = dz @ parts[2 * li].T
            Rdx = Rdz @ parts[2 * li].T + dz @ vparts[2 * li].T
            d1 = act_d(spec.hidden_act, acts[li])
            Rdz = Rdx * d1 + dx * act_dd_over_d(spec.hidden_act, acts[li]) * Racts[li]
            dz = dx * d1
    out[2 * nl] = (Rds * s_mask).reshape(1, -1)
    return spec.flatten(out)


def adapt(spec, thetas_tasks, slabs, step_sizes, kind=INNER_RATIO):
    """MAMLAlgo._adapt (meta_algos/base.py:217-242): theta'_i = theta_i - alpha * grad L_i(theta_i),
    explicit per-task parameters => raw (unclipped) log_std (gaussian_mlp_policy.py:182)."""
    out = []
    for th, slab in zip(thetas_tasks, slabs):
        g = loss_and_grad(spec, th, slab, kind, clip_log_std=False)['grad']
        out.append(np.asarray(th, dtype=np.float64) - step_sizes * g)
    return out


def meta_objective_and_grad(spec, theta, all_slabs, step_sizes, inner_kl_coeff, clip_eps,
                            inner_kind=INNER_RATIO, outer_kind=OUTER_CLIP, want_grad=True, tasks=None,
                            n_tasks_total=None):
    """ProMP.build_graph (pro_mp.py:67-163) evaluated at theta, with the exact gradient.

    all_slabs: list[K+1] of list[M] of slab dicts.  Returns dict(loss, inner_kl [K], outer_kl, grad,
    adapted [M] thetas).  ``tasks``/``n_tasks_total`` restrict the *sum* to a shard of tasks while
    keeping the 1/M of the full meta-batch (used by the multi-process sharding test): the returned
    quantities are then partial sums that add up across shards.
    """
    K = len(all_slabs) - 1
    assert K >= 1
    M = len(all_slabs[0])
    task_ids = range(M) if tasks is None else tasks
    Mtot = M if n_tasks_total is None else n_tasks_total
    theta = np.asarray(theta, dtype=np.float64)
    eta = np.asarray(inner_kl_coeff, dtype=np.float64)
    loss_sum, okl_sum, ikl_sum = 0.0, 0.0, np.zeros(K)
    grad_sum = np.zeros_like(theta)
    adapted = []
    for i in task_ids:
        thetas, gkls = [theta], []
        for k in range(K):
            r = loss_and_grad(spec, thetas[k], all_slabs[k][i], inner_kind, clip_log_std=(k == 0))
            ikl_sum[k] += r['kl']
            gkls.append(r['grad_kl'])
            thetas.append(thetas[k] - step_sizes * r['grad'])                       # base.py:206-211
        r = loss_and_grad(spec, thetas[K], all_slabs[K][i], outer_kind, clip_log_std=False,
                          clip_eps=clip_eps, want_grad=want_grad)
        loss_sum += r['loss']
        okl_sum += r['kl']
        adapted.append(thetas[K])
        if want_grad:
            lam = r['grad']
            for k in range(K - 1, -1, -1):
                lam = lam - hvp(spec, thetas[k], all_slabs[k][i], step_sizes * lam, inner_kind,
                                clip_log_std=(k == 0)) + (eta[k] / K) * gkls[k]
            grad_sum += lam
    inner_kl = ikl_sum / Mtot                                                        # pro_mp.py:122
    out = dict(loss=loss_sum / Mtot + float(np.mean(eta * inner_kl)),               # pro_mp.py:151-155
               inner_kl=inner_kl, outer_kl=okl_sum / Mtot, adapted=adapted)
    if want_grad:
        out['grad'] = grad_sum / Mtot
    return out


class AdamState:
    """tf.train.AdamOptimizer slots (m, v) and step count; persists across iterations."""
    def __init__(self, n):
        self.m = np.zeros(n)
        self.v = np.zeros(n)
        self.t = 0


def adam_step(theta, grad, st, lr, b1=0.9, b2=0.999, eps=1e-8):
    st.t += 1
    lr_t = lr * np.sqrt(1.0 - b2 ** st.t) / (1.0 - b1 ** st.t)
    st.m = b1 * st.m + (1 - b1) * grad
    st.v = b2 * st.v + (1 - b2) * grad * grad
    return theta - lr_t * st.m / (np.sqrt(st.v) + eps)


def adapt_kl_coeff(kl_coeff, kl_values, kl_target):
    """pro_mp.py:201-214."""
    out = []
    for c, klv in zip(kl_coeff, kl_values):
        if klv < kl_target / 1.5:
            c = c / 2
        elif klv > kl_target * 1.5:
            c = c * 2
        out.append(c)
    return np.array(out)


def optimize_policy(spec, theta, all_slabs, step_sizes, inner_kl_coeff, clip_eps, adam, lr, num_epochs,
                    inner_kind=INNER_RATIO, outer_kind=OUTER_CLIP):
    """ProMP.optimize_policy without logging (pro_mp.py:165-199):
    E full-batch Adam epochs (loss of the first epoch = 'LossBefore'), then compute_stats."""
    loss_before = None
    for _ in range(num_epochs):
        r = meta_objective_and_grad(spec, theta, all_slabs, step_sizes, inner_kl_coeff, clip_eps,
                                    inner_kind, outer_kind)
        if loss_before is None:
            loss_before = r['loss']
        theta = adam_step(theta, r['grad'], adam, lr)
    r = meta_objective_and_grad(spec, theta, all_slabs, step_sizes, inner_kl_coeff, clip_eps,
                                inner_kind, outer_kind, want_grad=False)
    return theta, dict(loss_before=loss_before, loss_after=r['loss'], inner_kl=r['inner_kl'],
                       outer_kl=r['outer_kl'])
