"""ConjugateGradientOptimizer + FiniteDifferenceHvp (reference:
meta_policy_search/optimizers/conjugate_gradient_optimizer.py:8-354).

Host-side control logic exactly as in the reference (CG, initial step from the quadratic model, backtracking line
search, step rejection); every evaluation it asks for -- loss, constraint value, loss gradient, constraint gradient --
is one pass of the device kernels through an ``evaluator`` object (see meta_algos/trpo_maml.py), including the
all-reduce over ranks when a communicator is attached.
"""
import numpy as np

from ..utils import logger


class FiniteDifferenceHvp(object):
    """Hx ~ (grad_c(theta + eps x) - grad_c(theta - eps x)) / (2 eps)   (conjugate_gradient_optimizer.py:59-89)"""

    def __init__(self, base_eps=1e-5, symmetric=True, grad_clip=None):
        self.base_eps = np.float32(base_eps)
        self.symmetric = symmetric
        self.grad_clip = grad_clip
        self.reg_coeff = None
        self._ev = None

    def build_graph(self, evaluator, reg_coeff):
        self._ev = evaluator
        self.reg_coeff = reg_coeff

    def constraint_gradient(self):
        return self._ev.constraint_gradient()

    def Hx(self, x):
        assert isinstance(x, np.ndarray)
        ev = self._ev
        theta = ev.get_theta().copy()
        eps = self.base_eps
        ev.set_theta(theta + eps * x)
        g_plus = ev.constraint_gradient()
        ev.set_theta(theta)
        if self.symmetric:
            ev.set_theta(theta - eps * x)
            g_minus = ev.constraint_gradient()
            ev.set_theta(theta)
            return (g_plus - g_minus) / (2 * eps)
        g = ev.constraint_gradient()
        return (g_plus - g) / eps

    def build_eval(self):
        def evaluate_hessian(x):
            return self.Hx(x) + self.reg_coeff * x
        return evaluate_hessian


class ConjugateGradientOptimizer(object):
    """Args as the reference (conjugate_gradient_optimizer.py:107-148)."""

    def __init__(self, cg_iters=10, reg_coeff=0, subsample_factor=1., backtrack_ratio=0.8, max_backtracks=15,
                 debug_nan=False, accept_violation=False, hvp_approach=None):
        self._cg_iters = cg_iters
        self._reg_coeff = reg_coeff
        self._subsample_factor = subsample_factor
        self._backtrack_ratio = backtrack_ratio
        self._max_backtracks = max_backtracks
        self._max_constraint_val = None
        self._constraint_name = 'kl-div'
        self._debug_nan = debug_nan
        self._accept_violation = accept_violation
        self._hvp_approach = hvp_approach if hvp_approach is not None else FiniteDifferenceHvp()
        self._ev = None

    def build_graph(self, evaluator, leq_constraint_value):
        """evaluator: object with loss(), constraint_val(), gradient(), constraint_gradient(), get_theta(), set_theta()"""
        self._ev = evaluator
        self._max_constraint_val = leq_constraint_value
        self._hvp_approach.build_graph(evaluator, self._reg_coeff)

    def loss(self, *_):
        return self._ev.loss()

    def constraint_val(self, *_):
        return self._ev.constraint_val()

    def gradient(self, *_):
        return self._ev.gradient()

    def optimize(self, *_):
        """conjugate_gradient_optimizer.py:239-307"""
        ev = self._ev
        logger.log('Start CG optimization')
        loss_before = self.loss()
        gradient = self.gradient()
        Hx = self._hvp_approach.build_eval()
        descent_direction = conjugate_gradients(Hx, gradient, cg_iters=self._cg_iters)
        initial_step_size = np.sqrt(2.0 * self._max_constraint_val *
                                    (1. / (descent_direction.dot(Hx(descent_direction)) + 1e-8)))
        if np.isnan(initial_step_size):
            logger.log('Initial step size is NaN! Rejecting the step!')
            self.last = dict(loss_before=loss_before, n_backtracks=0, rejected=True, descent_direction=descent_direction,
                             initial_step_size=float('nan'))
            return
        initial_descent_step = initial_step_size * descent_direction
        prev = ev.get_theta().copy()
        loss, constraint_val, n_iter, violated = 0, 0, 0, False
        for n_iter, ratio in enumerate(self._backtrack_ratio ** np.arange(self._max_backtracks)):
            ev.set_theta(prev - ratio * initial_descent_step)
            loss, constraint_val = self.loss(), self.constraint_val()
            if loss < loss_before and constraint_val <= self._max_constraint_val:
                break
        if np.isnan(loss):
            violated = True
            logger.log('Line search violated because loss is NaN')
        if np.isnan(constraint_val):
            violated = True
            logger.log('Line search violated because constraint %s is NaN' % self._constraint_name)
        if loss >= loss_before:
            violated = True
            logger.log('Line search violated because loss not improving')
        if constraint_val >= self._max_constraint_val:
            violated = True
            logger.log('Line search violated because constraint %s is violated' % self._constraint_name)
        if violated and not self._accept_violation:
            logger.log('Line search condition violated. Rejecting the step!')
            ev.set_theta(prev)
        logger.log('backtrack iters: %d' % n_iter)
        self.last = dict(loss_before=loss_before, n_backtracks=n_iter, rejected=bool(violated and not self._accept_violation),
                         descent_direction=descent_direction, initial_step_size=float(initial_step_size))


def conjugate_gradients(f_Ax, b, cg_iters=10, verbose=False, residual_tol=1e-10):
    """Demmel p 312 (conjugate_gradient_optimizer.py:325-354)"""
    p = b.copy()
    r = b.copy()
    x = np.zeros_like(b, dtype=np.float32)
    rdotr = r.dot(r)
    for i in range(cg_iters):
        z = f_Ax(p)
        v = rdotr / p.dot(z)
        x += v * p
        r -= v * z
        newrdotr = r.dot(r)
        mu = newrdotr / rdotr
        p = r + mu * p
        rdotr = newrdotr
        if rdotr < residual_tol:
            break
    return x
