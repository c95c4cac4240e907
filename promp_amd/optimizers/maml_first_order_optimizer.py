"""First-order optimisers of the MAML outer step: MAMLFirstOrderOptimizer, MAMLPPOOptimizer.

Behaviour contract (reference: meta_policy_search/optimizers/maml_first_order_optimizer.py:5-163; what ProMP and VPG-MAML call
through `self.optimizer`):

  * `optimize(inputs)` takes `max_epochs` full-batch steps of tf.train.AdamOptimizer(learning_rate) on the meta-objective and returns
    the loss BEFORE the first step (:82-115; `num_minibatches` and `tolerance` are accepted and unused, as there, :41, :97-113);
  * `loss(inputs)` evaluates the meta-objective at the current parameters (:66-80);
  * MAMLPPOOptimizer.compute_stats(inputs) returns (loss, inner_kl [per inner step], outer_kl) at the current parameters (:146-163).

There is no graph to build here.  `build_graph` binds the optimiser to the algorithm plugin whose objective it steps (the reference
binds a loss tensor, a target policy and placeholders); every call is then one or more passes of the device kernels through the
plugin's DeviceSession: promp_optimize runs the E epochs AND the statistics pass behind them in one call, so `optimize` keeps what
the device handed back and a `compute_stats` / `loss` that follows it at unchanged parameters, data and coefficients answers from
there instead of walking the batch again (ProMP.optimize_policy asks exactly in that order, pro_mp.py:185-190).
"""
import numpy as np

from .. import _lib


class Optimizer(object):
    """optimizers/base.py:3-36: the three calls an optimiser answers"""

    def build_graph(self, loss, target=None, input_ph_dict=None):
        raise NotImplementedError

    def optimize(self, input_val_dict=None):
        raise NotImplementedError

    def loss(self, input_val_dict=None):
        raise NotImplementedError


class MAMLFirstOrderOptimizer(Optimizer):
    """
    Args (maml_first_order_optimizer.py:22-46): tf_optimizer_cls / tf_optimizer_args -- only Adam exists on the device; the class
    argument is accepted for call compatibility and `tf_optimizer_args['learning_rate']` is honoured --, learning_rate, max_epochs,
    tolerance (unused), num_minibatches (unused), verbose.
    """

    def __init__(self, tf_optimizer_cls=None, tf_optimizer_args=None, learning_rate=1e-3, max_epochs=1, tolerance=1e-6,
                 num_minibatches=1, verbose=False):
        args = dict(tf_optimizer_args or {})
        self._learning_rate = float(args.get('learning_rate', learning_rate))
        self._max_epochs = int(max_epochs)
        self._tolerance = tolerance
        self._num_minibatches = num_minibatches      # unused, as in the reference
        self._verbose = verbose
        self._algo = None
        self._memo = None                            # (key, result of the last promp_optimize)

    def build_graph(self, loss, target=None, input_ph_dict=None, **_):
        """`loss`: the algorithm plugin (a MAMLAlgo with a DeviceSession: .session, .inner_kind, .outer_kind and -- ProMP --
        .clip_eps, .inner_kl_coeff).  `target` / `input_ph_dict` are the reference's policy and placeholders: nothing to bind."""
        assert hasattr(loss, 'session'), 'build_graph binds the optimiser to an algorithm plugin (promp_amd.meta_algos.*)'
        self._algo = loss
        self._memo = None

    # what the device evaluates with: the algorithm's current settings unless the call's dict overrides them (the reference feeds
    # clip_eps and the KL coefficients through the same dict as the samples, pro_mp.py:176-183)
    def _settings(self, input_val_dict):
        a, d = self._algo, (input_val_dict if isinstance(input_val_dict, dict) else {})
        K = a.num_inner_grad_steps
        clip_eps = float(d.get('clip_eps', getattr(a, 'clip_eps', 0.0)))
        eta = np.asarray(d.get('inner_kl_coeff', getattr(a, 'inner_kl_coeff', np.zeros(K))), dtype=np.float32).reshape(-1)
        return clip_eps, eta

    def _key(self, clip_eps, eta):
        s = self._algo.session
        return (s.param_version, tuple(s.upload_serial), clip_eps, eta.tobytes(), self._algo.inner_kind, self._algo.outer_kind)

    def _place(self, input_val_dict):
        if isinstance(input_val_dict, (list, tuple)):          # all_samples_data: sampling step k into slot k
            self._algo._place_steps(input_val_dict)

    def optimize(self, input_val_dict=None):
        assert self._algo is not None, 'build_graph first'
        self._place(input_val_dict)
        clip_eps, eta = self._settings(input_val_dict)
        a = self._algo
        res = a.session.optimize(self._max_epochs, self._learning_rate, clip_eps, eta, a.inner_kind, a.outer_kind)
        a.session.param_version += 1
        self._memo = (self._key(clip_eps, eta), res)
        return res['loss_before']

    def _evaluate(self, input_val_dict):
        assert self._algo is not None, 'build_graph first'
        self._place(input_val_dict)
        clip_eps, eta = self._settings(input_val_dict)
        key = self._key(clip_eps, eta)
        if self._memo is not None and self._memo[0] == key:
            r = self._memo[1]
            return dict(loss=r['loss_after'], inner_kl=r['inner_kl'], outer_kl=r['outer_kl'])
        a = self._algo
        r = a.session.meta_eval(clip_eps, eta, a.inner_kind, a.outer_kind)
        self._memo = (key, dict(loss_before=r['loss'], loss_after=r['loss'], inner_kl=r['inner_kl'], outer_kl=r['outer_kl']))
        return r

    def loss(self, input_val_dict=None):
        return self._evaluate(input_val_dict)['loss']

    @property
    def last_result(self):
        """what the last promp_optimize / evaluation returned: loss_before, loss_after, inner_kl, outer_kl"""
        return None if self._memo is None else self._memo[1]


class MAMLPPOOptimizer(MAMLFirstOrderOptimizer):
    """Adds the statistics of the PPO-style outer objective (maml_first_order_optimizer.py:118-163)"""

    def build_graph(self, loss, target=None, input_ph_dict=None, inner_kl=None, outer_kl=None, **_):
        super(MAMLPPOOptimizer, self).build_graph(loss, target, input_ph_dict)

    def compute_stats(self, input_val_dict=None):
        r = self._evaluate(input_val_dict)
        return r['loss'], r['inner_kl'], r['outer_kl']


__all__ = ['Optimizer', 'MAMLFirstOrderOptimizer', 'MAMLPPOOptimizer']
_ = _lib      # (the kinds the plugins carry are _lib's INNER_* / OUTER_* constants)
