"""Trust-region step for TRPO-MAML: ConjugateGradientOptimizer + FiniteDifferenceHvp.

Behaviour contract (what meta_algos/trpo_maml.py and the parity tests rely on; the reference implements the same
contract in meta_policy_search/optimizers/conjugate_gradient_optimizer.py:8-104 (FiniteDifferenceHvp), 107-307 (ConjugateGradientOptimizer),
325-354 (conjugate_gradients)):

  * the search direction d solves  (H + reg I) d = g  approximately: `cg_iters` conjugate-gradient iterations on products
    H x only, H = Hessian of the constraint (mean KL), g = gradient of the loss;
  * H x is a finite difference of the constraint gradient, symmetric by default with eps = 1e-5;
  * the full step is  sqrt(2 delta / (d^T H d + 1e-8)) * d  (the KL ball of radius delta under the quadratic model);
    a NaN step length rejects the update outright;
  * backtracking: scales ratio^0, ratio^1, ... (at most `max_backtracks`); the first candidate with a lower loss and a
    constraint value <= delta ends the search;
  * the LAST candidate tried is then audited: a NaN loss or constraint, a loss that did not improve, or a constraint value
    >= delta (note: >=, while the search accepts <=) marks the step as violated, and unless `accept_violation` the
    parameters are restored.

Every quantity the step needs -- loss, constraint value, loss gradient, constraint gradient -- is one pass of the device
kernels, requested through an `evaluator` object (meta_algos/trpo_maml.py), which also hides the all-reduce over ranks
when a communicator is attached.  This module is pure host-side control flow.
"""
import numpy as np

from ..utils import logger


def conjugate_gradients(f_Ax, b, cg_iters=10, verbose=False, residual_tol=1e-10):
    """Approximate solution of A x = b for a symmetric positive definite A known only through x -> A x.

    Plain conjugate gradients started at x = 0; the iterate is kept in float32 (the parameters it is added to are
    float32), search direction and residual keep b's dtype.  Stops after `cg_iters` products or once the squared
    residual norm drops below `residual_tol`."""
    solution = np.zeros(np.shape(b), dtype=np.float32)
    residual = np.array(b, copy=True)
    direction = np.array(b, copy=True)
    res_sq = residual.dot(residual)
    for it in range(cg_iters):
        curved = f_Ax(direction)
        step = res_sq / direction.dot(curved)
        solution += step * direction
        residual -= step * curved
        new_res_sq = residual.dot(residual)
        if verbose:
            logger.log('cg iteration %d: |r|^2 = %.3e' % (it, float(new_res_sq)))
        direction = residual + (new_res_sq / res_sq) * direction
        res_sq = new_res_sq
        if res_sq < residual_tol:
            break
    return solution


class FiniteDifferenceHvp(object):
    """Hessian-vector products of the constraint from gradients at displaced parameters:
        symmetric:  H x ~ (grad c(theta + eps x) - grad c(theta - eps x)) / (2 eps)
        one-sided:  H x ~ (grad c(theta + eps x) - grad c(theta)) / eps
    The parameters are back at theta when a product returns."""

    def __init__(self, base_eps=1e-5, symmetric=True, grad_clip=None):
        self.base_eps = np.float32(base_eps)
        self.symmetric = symmetric
        self.grad_clip = grad_clip
        self.reg_coeff = None
        self._ev = None

    def build_graph(self, evaluator, reg_coeff):
        self._ev, self.reg_coeff = evaluator, reg_coeff

    def constraint_gradient(self):
        return self._ev.constraint_gradient()

    def _gradient_at(self, theta):
        self._ev.set_theta(theta)
        return self._ev.constraint_gradient()

    def Hx(self, x):
        assert isinstance(x, np.ndarray)
        centre = np.array(self._ev.get_theta(), copy=True)
        eps = self.base_eps
        ahead = self._gradient_at(centre + eps * x)
        if self.symmetric:
            behind = self._gradient_at(centre - eps * x)
            self._ev.set_theta(centre)
            return (ahead - behind) / (2 * eps)
        here = self._gradient_at(centre)
        return (ahead - here) / eps

    def build_eval(self):
        """x -> (H + reg I) x"""
        return lambda x: self.Hx(x) + self.reg_coeff * x


class ExactDeviceHvp(object):
    """Hessian-vector products of the constraint computed exactly on the device (promp_constraint_hvp: 2K+1 R-operator
    passes, J^T H_KL J through the K inner steps) instead of two displaced constraint gradients per product.  Not in the
    reference, which only has the finite-difference approach (SURVEY 8f row 2 asks for this one); FiniteDifferenceHvp stays
    the default and the parity mode.  Exact where the constraint is evaluated by TRPO: at the parameters the last step's
    samples were drawn with (there the constraint's gradient wrt the adapted parameters is zero, and with it every term
    that differentiates the adaptation Jacobians)."""

    def __init__(self, allow_fallback=True, fallback_eps=1e-5, fallback_symmetric=True):
        """allow_fallback: where the exact product does not exist (below) use a FiniteDifferenceHvp(fallback_eps,
        fallback_symmetric) instead -- said once in the log --; False: raise there"""
        self.reg_coeff = None
        self._ev = None
        self._fresh = False
        self._fd = None
        self._allow_fallback, self._fd_args, self._told = bool(allow_fallback), (fallback_eps, fallback_symmetric), False

    def build_graph(self, evaluator, reg_coeff):
        self._ev, self.reg_coeff = evaluator, reg_coeff
        self._fd = FiniteDifferenceHvp(base_eps=self._fd_args[0], symmetric=self._fd_args[1])
        self._fd.build_graph(evaluator, reg_coeff)

    def constraint_gradient(self):
        return self._ev.constraint_gradient()

    def Hx(self, x):
        assert isinstance(x, np.ndarray)
        if getattr(self._ev, 'exact_hvp_available', lambda: True)() is False:
            # ranks that exchange through the session's `collective` hold no communicator: the library refuses to present one
            # shard's product as the batch's.  The reference's own construction works there (its gradients cross the collective).
            if not self._allow_fallback:
                raise RuntimeError('ExactDeviceHvp: the exact constraint product needs the library\'s communicator; this session '
                                   'exchanges through `collective` (allow_fallback=True takes finite differences there)')
            if not self._told:
                self._told = True
                logger.log('ExactDeviceHvp: no exact constraint product on ranks that exchange through `collective`; using finite '
                           'differences (eps %g, %s)'
                           % (self._fd_args[0], 'symmetric: two gradient evaluations per product' if self._fd_args[1] else 'one-sided'))
            return self._fd.Hx(x)
        out = self._ev.constraint_hvp(x, refresh_chain=not self._fresh)   # the adapted parameters are computed once per theta
        self._fresh = True
        return out

    def build_eval(self):
        """x -> (H + reg I) x at the evaluator's current parameters"""
        self._fresh = False
        return lambda x: self.Hx(x) + self.reg_coeff * x


class ConjugateGradientOptimizer(object):
    """
    Args: cg_iters=10, reg_coeff=0 (Tikhonov term added to H), subsample_factor=1. (kept for signature compatibility:
    the device evaluates the constraint on the whole batch), backtrack_ratio=0.8, max_backtracks=15, debug_nan=False,
    accept_violation=False, hvp_approach=None (a FiniteDifferenceHvp by default)

    After optimize(), `last` holds loss_before, descent_direction, initial_step_size, n_backtracks and rejected.
    """

    def __init__(self, cg_iters=10, reg_coeff=0, subsample_factor=1., backtrack_ratio=0.8, max_backtracks=15,
                 debug_nan=False, accept_violation=False, hvp_approach=None, device_solve=True):
        self._device_solve = bool(device_solve)
        self._cg_iters = cg_iters
        self._reg_coeff = reg_coeff
        self._subsample_factor = subsample_factor
        self._backtrack_ratio = backtrack_ratio
        self._max_backtracks = max_backtracks
        self._debug_nan = debug_nan
        self._accept_violation = accept_violation
        self._hvp_approach = FiniteDifferenceHvp() if hvp_approach is None else hvp_approach
        self._constraint_name = 'kl-div'
        self._max_constraint_val = None
        self._ev = None
        self.last = None

    def build_graph(self, evaluator, leq_constraint_value):
        """evaluator: loss(), constraint_val(), gradient(), constraint_gradient(), get_theta(), set_theta(theta)"""
        self._ev = evaluator
        self._max_constraint_val = leq_constraint_value
        self._hvp_approach.build_graph(evaluator, self._reg_coeff)

    def loss(self, *_):
        return self._ev.loss()

    def constraint_val(self, *_):
        return self._ev.constraint_val()

    def gradient(self, *_):
        return self._ev.gradient()

    # ---- the solve ----
    def _solve_on_device(self, grad):
        """The reference's loop -- cg_iters products, each two displaced constraint gradients, then the closing product -- asks the
        device for 2 cg_iters + 2 gradients one at a time, and every one of them crosses to the host and back.  With the evaluator
        of the device kernels and one of this module's own product constructions the library runs the same loop with the products
        enqueued back to back (promp_cg_solve: same float32 vectors, dot products in float64).  None: not available, loop here."""
        solve = getattr(self._ev, 'cg_solve', None)
        hv = self._hvp_approach
        if not self._device_solve or solve is None:
            return None
        if type(hv) is FiniteDifferenceHvp and hv.grad_clip is None:
            mode, eps = (0 if hv.symmetric else 1), float(hv.base_eps)
        elif type(hv) is ExactDeviceHvp and getattr(self._ev, 'exact_hvp_available', lambda: True)() is not False:
            mode, eps = 2, 1e-5
        else:
            return None
        return solve(grad, self._cg_iters, float(self._reg_coeff), mode, eps=eps)

    # ---- the step ----
    def _search(self, origin, full_step, loss_before):
        """backtracking over the scales ratio^k; returns (loss, constraint, k) of the last candidate tried"""
        delta = self._max_constraint_val
        loss = constraint = 0
        k = 0
        for k in range(self._max_backtracks):
            self._ev.set_theta(origin - (self._backtrack_ratio ** k) * full_step)
            loss, constraint = self.loss(), self.constraint_val()
            if loss < loss_before and constraint <= delta:
                break
        return loss, constraint, k

    def _audit(self, loss, constraint, loss_before):
        """reasons why the candidate must not be kept (empty list: keep it)"""
        name, delta = self._constraint_name, self._max_constraint_val
        checks = ((np.isnan(loss), 'the loss is NaN'),
                  (np.isnan(constraint), 'the constraint %s is NaN' % name),
                  (loss >= loss_before, 'the loss did not improve'),
                  (constraint >= delta, 'the constraint %s reached its bound' % name))
        return [why for failed, why in checks if failed]

    def optimize(self, *_):
        ev = self._ev
        logger.log('trust-region step: conjugate gradients')
        loss_before = self.loss()
        grad = self.gradient()
        solved = self._solve_on_device(grad)
        if solved is not None:
            direction, curved = solved
        else:
            curvature = self._hvp_approach.build_eval()
            direction = conjugate_gradients(curvature, grad, cg_iters=self._cg_iters)
            curved = direction.dot(curvature(direction))
        with np.errstate(invalid='ignore'):       # negative curvature along the direction: NaN, rejected below (as the reference, :264-268)
            length = np.sqrt(2.0 * self._max_constraint_val * (1. / (curved + 1e-8)))
        self.last = dict(loss_before=loss_before, gradient=grad, descent_direction=direction, initial_step_size=float(length),
                         n_backtracks=0, rejected=False)
        if np.isnan(length):
            logger.log('trust-region step: step length is NaN, update rejected')
            self.last['rejected'] = True
            return
        origin = np.array(ev.get_theta(), copy=True)
        loss, constraint, tried = self._search(origin, length * direction, loss_before)
        problems = self._audit(loss, constraint, loss_before)
        for why in problems:
            logger.log('trust-region step: line search failed, ' + why)
        rejected = bool(problems) and not self._accept_violation
        if rejected:
            logger.log('trust-region step: update rejected, parameters restored')
            ev.set_theta(origin)
        logger.log('trust-region step: %d backtracking step(s)' % tried)
        self.last.update(n_backtracks=tried, rejected=rejected)
