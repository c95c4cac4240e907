"""Oracle for reference rows a8-a9: tanh-MLP Gaussian policy math in float64.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates (paths relative to /root/reference/meta_policy_search/):
  * policies/networks/mlp.py:65-119         forward_mlp  (x @ W + b, tanh hidden, identity output)
  * policies/gaussian_mlp_policy.py:49-80   parameter set + names; log_std = max(var, log(min_std))
  * policies/gaussian_mlp_policy.py:142-184 distribution_info_sym (clipped log_std from shared
                                            variables, raw log_std from explicit params)
  * policies/distributions/diagonal_gaussian.py:16-45, 71-109  kl / likelihood ratio / log-lik

Flat parameter vector layout == the reference's OrderedDict insertion order
(policies/base.py:271-277, gaussian_mlp_policy.py:78-80):
  mean_network/hidden_0/kernel [O,H1] row-major, .../hidden_0/bias [H1], hidden_1/kernel, hidden_1/bias,
  ..., mean_network/output/kernel [Hl,A], .../output/bias [A], log_std_network/log_std_var [1,A]
"""
from collections import OrderedDict
import numpy as np

LOG_2PI = float(np.log(2.0 * np.pi))


class PolicySpec:
    def __init__(self, obs_dim, action_dim, hidden_sizes=(64, 64), min_std=1e-6, hidden_act='tanh', output_act='identity'):
        assert hidden_act in ('tanh', 'relu', 'identity')      # policies/networks/mlp.py:47 (None = identity)
        assert output_act in ('tanh', 'relu', 'identity')      # policies/networks/mlp.py:53-60, 114-117: output_nonlinearity on the mean (None = identity)
        self.hidden_act = hidden_act
        self.output_act = output_act
        self.obs_dim = int(obs_dim)
        self.action_dim = int(action_dim)
        self.hidden_sizes = tuple(int(h) for h in hidden_sizes)
        self.min_log_std = float(np.log(min_std))
        sizes = (self.obs_dim,) + self.hidden_sizes + (self.action_dim,)
        self.layer_shapes = [(sizes[i], sizes[i + 1]) for i in range(len(sizes) - 1)]
        self.names, self.shapes = [], []
        for li, (fi, fo) in enumerate(self.layer_shapes):
            lname = 'output' if li == len(self.layer_shapes) - 1 else 'hidden_%d' % li
            self.names += ['mean_network/%s/kernel' % lname, 'mean_network/%s/bias' % lname]
            self.shapes += [(fi, fo), (fo,)]
        self.names.append('log_std_network/log_std_var')
        self.shapes.append((1, self.action_dim))
        self.sizes = [int(np.prod(s)) for s in self.shapes]
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(int)
        self.n_params = int(self.offsets[-1])

    # ---- flat <-> structured -------------------------------------------------
    def unflatten(self, theta):
        theta = np.asarray(theta)
        return [theta[self.offsets[i]:self.offsets[i + 1]].reshape(self.shapes[i]) for i in range(len(self.shapes))]

    def flatten(self, parts):
        return np.concatenate([np.asarray(p).reshape(-1) for p in parts])

    def to_ordered_dict(self, theta):
        return OrderedDict(zip(self.names, self.unflatten(theta)))

    def from_ordered_dict(self, d):
        assert list(d.keys()) == self.names, "parameter keys must match with variable"
        return self.flatten([d[k] for k in self.names])

    def log_std_slice(self):
        return slice(self.offsets[-2], self.offsets[-1])

    def init_params(self, rng, init_std=1.0):
        """Xavier-uniform kernels, zero biases, log_std = log(init_std).
        mlp.py:12-13, gaussian_mlp_policy.py:37,63-69."""
        parts = []
        for (fi, fo) in self.layer_shapes:
            lim = np.sqrt(6.0 / (fi + fo))
            parts += [rng.uniform(-lim, lim, size=(fi, fo)), np.zeros(fo)]
        parts.append(np.full((1, self.action_dim), np.log(init_std)))
        return self.flatten(parts).astype(np.float64)


def act_f(kind, z):
    """hidden nonlinearity (mlp.py:47): tanh (default, policies/base.py:31), relu, identity (hidden_nonlinearity=None)"""
    return np.tanh(z) if kind == 'tanh' else np.maximum(z, 0.0) if kind == 'relu' else z


def act_d(kind, h):
    """its derivative as a function of the OUTPUT h (TF: relu'(0) = 0)"""
    return 1.0 - h ** 2 if kind == 'tanh' else (h > 0).astype(np.float64) if kind == 'relu' else np.ones_like(h)


def act_dd_over_d(kind, h):
    """f''(z) / f'(z) as a function of h: the R-operator of f'(z) is this times R{h} (tanh: -2 h; relu / identity: 0)"""
    return -2.0 * h if kind == 'tanh' else np.zeros_like(h)


def forward(spec, theta, obs, clip_log_std):
    """-> (mean [N,A], log_std [A], cache).  mlp.py:65-119, gaussian_mlp_policy.py:142-184."""
    parts = spec.unflatten(np.asarray(theta, dtype=np.float64))
    x = np.asarray(obs, dtype=np.float64)
    acts = [x]
    nl = len(spec.layer_shapes)
    for li in range(nl):
        z = x @ parts[2 * li] + parts[2 * li + 1]
        x = act_f(spec.hidden_act, z) if li < nl - 1 else act_f(spec.output_act, z)      # mlp.py:114-117
        acts.append(x)
    s_raw = parts[-1].reshape(-1)
    if clip_log_std:
        s = np.maximum(s_raw, spec.min_log_std)            # gaussian_mlp_policy.py:71,163
        s_mask = (s_raw >= spec.min_log_std).astype(np.float64)   # d max(x,c)/dx, TF convention x>=c
    else:
        s = s_raw                                          # gaussian_mlp_policy.py:182
        s_mask = np.ones_like(s_raw)
    return acts[-1], s, dict(acts=acts, parts=parts, s_mask=s_mask)


def log_likelihood(actions, mean, log_std):
    """diagonal_gaussian.py:89-109."""
    zs = (actions - mean) / np.exp(log_std)
    return -np.sum(log_std * np.ones_like(mean), axis=-1) - 0.5 * np.sum(zs ** 2, axis=-1) \
        - 0.5 * mean.shape[-1] * LOG_2PI


def likelihood_ratio(actions, old_mean, old_log_std, mean, log_std):
    """diagonal_gaussian.py:71-87."""
    return np.exp(log_likelihood(actions, mean, log_std) - log_likelihood(actions, old_mean, old_log_std))


def kl(old_mean, old_log_std, mean, log_std):
    """KL(old || new) per row.  diagonal_gaussian.py:16-45 (note the +1e-8)."""
    old_std, new_std = np.exp(old_log_std), np.exp(log_std)
    num = (old_mean - mean) ** 2 + old_std ** 2 - new_std ** 2
    den = 2 * new_std ** 2 + 1e-8
    return np.sum(num / den + log_std - old_log_std, axis=-1)
