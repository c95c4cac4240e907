"""Oracle for reference row a15: the TRPO outer step of MAML-TRPO, float64 NumPy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates
  * meta_algos/trpo_maml.py:69-191                       objective, constraint, optimize_policy
  * optimizers/conjugate_gradient_optimizer.py:59-89      symmetric finite-difference HVP of the constraint, eps = 1e-5
  * optimizers/conjugate_gradient_optimizer.py:239-307    step from the quadratic model, backtracking, rejection
  * optimizers/conjugate_gradient_optimizer.py:325-354    conjugate_gradients
on top of oracle/promp.py's meta-objective evaluation (outer kinds 'ratio' and 'kl').
"""
import numpy as np

from . import promp as pm


def conjugate_gradients(f_Ax, b, cg_iters=10, residual_tol=1e-10):
    p, r, x = b.copy(), b.copy(), np.zeros_like(b)
    rdotr = r.dot(r)
    for _ in range(cg_iters):
        z = f_Ax(p)
        v = rdotr / p.dot(z)
        x += v * p
        r -= v * z
        newrdotr = r.dot(r)
        p = r + (newrdotr / rdotr) * p
        rdotr = newrdotr
        if rdotr < residual_tol:
            break
    return x


def exploration_term(spec, theta, slabs0, coeffs, want_grad=True):
    """E-MAML (trpo_maml.py:137-144): mean_i [ -mean(adj_avg_rewards_i) * mean_n log pi_theta(a0_n | s0_n) ] and its gradient.
    coeffs[i] = mean(adj_avg_rewards of task i at the last sampling step); log_std clipped as in params=None graphs."""
    val, grad = 0.0, np.zeros(spec.n_params)
    for c, slab in zip(coeffs, slabs0):
        unit = dict(slab, advantages=np.ones_like(np.asarray(slab['advantages'], dtype=np.float64)))
        r = pm.loss_and_grad(spec, theta, unit, 'loglik', True)      # loss = -mean(log pi)
        val += c * r['loss']
        if want_grad:
            grad += c * r['grad']
    n = len(slabs0)
    return val / n, grad / n


def constraint_hvp_fd64(spec, theta, all_slabs, step_sizes, x, inner_kind='loglik', rel_eps=1e-6):
    """Hessian-vector product of the TRPO constraint (mean outer KL through the adaptation, trpo_maml.py:146-158) by a
    central difference of the float64 constraint gradient -- the reference's own construction
    (conjugate_gradient_optimizer.py:59-89) carried out in float64 with a step scaled to the direction, where its
    truncation + rounding error is ~1e-9 relative: the checker for the device's exact product."""
    theta = np.asarray(theta, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    K = len(all_slabs) - 1
    eps = rel_eps / max(np.linalg.norm(x), 1e-30) * max(np.linalg.norm(theta), 1.0)
    ev = lambda th: pm.meta_objective_and_grad(spec, th, all_slabs, step_sizes, np.zeros(K), 0.0, inner_kind, 'kl',
                                               want_grad=True)['grad']
    return (ev(theta + eps * x) - ev(theta - eps * x)) / (2 * eps)


def trpo_maml_step(spec, theta, all_slabs, step_sizes, inner_kind='loglik', max_kl=0.01, cg_iters=10, reg_coeff=0.0,
                   backtrack_ratio=0.8, max_backtracks=15, fd_eps=1e-5, explore_coeffs=None):
    K = len(all_slabs) - 1
    eta = np.zeros(K)
    theta = np.asarray(theta, dtype=np.float64)

    def ev(th, outer, grad):
        r = pm.meta_objective_and_grad(spec, th, all_slabs, step_sizes, eta, 0.0, inner_kind, outer, want_grad=grad)
        if explore_coeffs is not None and outer == 'ratio':
            v, g = exploration_term(spec, th, all_slabs[0], explore_coeffs, grad)
            r = dict(r, loss=r['loss'] + v)
            if grad:
                r['grad'] = r['grad'] + g
        return r

    r0 = ev(theta, 'ratio', True)
    loss_before, kl_before, g = r0['loss'], r0['outer_kl'], r0['grad']

    def Hx(x):
        gp = ev(theta + fd_eps * x, 'kl', True)['grad']
        gm = ev(theta - fd_eps * x, 'kl', True)['grad']
        return (gp - gm) / (2 * fd_eps) + reg_coeff * x

    d = conjugate_gradients(Hx, g, cg_iters)
    step0 = np.sqrt(2.0 * max_kl / (d.dot(Hx(d)) + 1e-8))
    out = dict(loss_before=loss_before, kl_before=kl_before, gradient=g, descent_direction=d, initial_step_size=step0)
    if np.isnan(step0):
        out.update(theta=theta, rejected=True, n_backtracks=0, loss_after=loss_before, kl_after=kl_before)
        return out
    loss = klv = 0.0
    n_iter = 0
    for n_iter, ratio in enumerate(backtrack_ratio ** np.arange(max_backtracks)):
        cand = theta - ratio * step0 * d
        r = ev(cand, 'ratio', False)
        loss, klv = r['loss'], r['outer_kl']
        if loss < loss_before and klv <= max_kl:
            break
    violated = bool(np.isnan(loss) or np.isnan(klv) or loss >= loss_before or klv >= max_kl)
    new_theta = theta if violated else cand
    r = ev(new_theta, 'ratio', False)
    out.update(theta=new_theta, rejected=violated, n_backtracks=n_iter, loss_after=r['loss'], kl_after=r['outer_kl'])
    return out
