"""Oracle for reference rows a1-a7: sample processing.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  float64 throughout, as
the reference (SciPy ``lfilter`` promotes to float64).

Every function cites the reference lines it restates (paths relative to
/root/reference/meta_policy_search/).
"""
import numpy as np

BASELINE_ZERO = 0
BASELINE_LINEAR_FEATURE = 1
BASELINE_LINEAR_TIME = 2


def discount_cumsum(x, discount):
    """y[t] = x[t] + discount * y[t+1], y[T] = 0.

    Restates utils/utils.py:74-81 (``lfilter([1],[1,-discount], x[::-1])[::-1]``)
    as the explicit recurrence the filter implements.
    """
    x = np.asarray(x, dtype=np.float64)
    y = np.zeros_like(x)
    run = np.zeros(x.shape[1:], dtype=np.float64)
    for t in range(x.shape[0] - 1, -1, -1):
        run = x[t] + discount * run
        y[t] = run
    return y


def linear_feature_baseline_features(obs):
    """[clip(o), clip(o)^2, t/100, (t/100)^2, (t/100)^3, 1]  ->  [T, 2*O+4].

    Restates baselines/linear_baseline.py:101-106.
    """
    obs = np.clip(np.asarray(obs), -10, 10)
    T = obs.shape[0]
    ts = np.arange(T).reshape(-1, 1) / 100.0
    return np.concatenate([obs, obs ** 2, ts, ts ** 2, ts ** 3, np.ones((T, 1))], axis=1)


def linear_time_baseline_features(obs):
    """[t/100, (t/100)^2, (t/100)^3, 1] -> [T, 4].  baselines/linear_baseline.py:122-126."""
    T = len(obs)
    ts = np.arange(T).reshape(-1, 1) / 100.0
    return np.concatenate([ts, ts ** 2, ts ** 3, np.ones((T, 1))], axis=1)


def _features(obs, baseline_kind):
    if baseline_kind == BASELINE_LINEAR_FEATURE:
        return linear_feature_baseline_features(obs)
    if baseline_kind == BASELINE_LINEAR_TIME:
        return linear_time_baseline_features(obs)
    raise ValueError(baseline_kind)


def fit_linear_baseline(path_obs, path_targets, baseline_kind=BASELINE_LINEAR_FEATURE, reg_coeff=1e-5):
    """Ridge normal equations solved with lstsq; NaN -> reg *= 10, at most 5 tries.

    Restates baselines/linear_baseline.py:55-77.
    Returns (coeffs [D], gram [D,D], rhs [D]).
    """
    featmat = np.concatenate([_features(o, baseline_kind) for o in path_obs], axis=0)
    target = np.concatenate([np.asarray(t, dtype=np.float64) for t in path_targets], axis=0)
    gram = featmat.T.dot(featmat)
    rhs = featmat.T.dot(target)
    reg = reg_coeff
    coeffs = None
    for _ in range(5):
        coeffs = np.linalg.lstsq(gram + reg * np.identity(featmat.shape[1]), rhs, rcond=-1)[0]
        if not np.any(np.isnan(coeffs)):
            break
        reg *= 10
    return coeffs, gram, rhs


def predict_linear_baseline(obs, coeffs, baseline_kind=BASELINE_LINEAR_FEATURE):
    """Phi . w, zeros when unfit.  baselines/linear_baseline.py:17-33."""
    if coeffs is None:
        return np.zeros(len(obs))
    return _features(obs, baseline_kind).dot(coeffs)


def compute_advantages(rewards, baselines, discount, gae_lambda):
    """delta[t] = r[t] + g*b[t+1] - b[t] (b[T]=0); adv = discount_cumsum(delta, g*lambda).

    Restates samplers/base.py:151-162.
    """
    b = np.append(np.asarray(baselines, dtype=np.float64), 0.0)
    deltas = np.asarray(rewards, dtype=np.float64) + discount * b[1:] - b[:-1]
    return discount_cumsum(deltas, discount * gae_lambda)


def normalize_advantages(adv):
    """(adv - mean) / (std_pop + 1e-8).  utils/utils.py:59-67."""
    return (adv - np.mean(adv)) / (adv.std() + 1e-8)


def shift_advantages_to_positive(adv):
    """adv - min + 1e-8.  utils/utils.py:70-71."""
    return (adv - np.min(adv)) + 1e-8


def compute_samples_data(paths, baseline_kind=BASELINE_LINEAR_FEATURE, discount=0.99, gae_lambda=1.0,
                         normalize_adv=False, positive_adv=False, reg_coeff=1e-5):
    """One task: returns -> baseline fit/predict -> GAE -> stack -> normalise.

    Restates samplers/base.py:99-133 (+ _stack_path_data :165-173).
    ``paths``: list of dicts with observations [T,O], actions [T,A], rewards [T]
    and optional agent_infos {mean, log_std}.  Like the reference this adds
    ``returns`` and ``advantages`` to every path dict.
    Returns (samples_data dict, coeffs or None).
    """
    for p in paths:
        p["returns"] = discount_cumsum(p["rewards"], discount)                    # base.py:103-104
    coeffs = None
    if baseline_kind != BASELINE_ZERO:
        coeffs, _, _ = fit_linear_baseline([p["observations"] for p in paths],
                                           [p["returns"] for p in paths],
                                           baseline_kind, reg_coeff)               # base.py:107
    for p in paths:
        if baseline_kind == BASELINE_ZERO:
            b = np.zeros(len(p["rewards"]))                                        # zero_baseline.py:41-54
        else:
            b = predict_linear_baseline(p["observations"], coeffs, baseline_kind)  # base.py:108
        p["advantages"] = compute_advantages(p["rewards"], b, discount, gae_lambda)  # base.py:111

    def cat(key):
        return np.concatenate([p[key] for p in paths])

    adv = cat("advantages")
    if normalize_adv:
        adv = normalize_advantages(adv)                                            # base.py:117-118
    if positive_adv:
        adv = shift_advantages_to_positive(adv)                                    # base.py:119-120
    out = dict(observations=cat("observations"), actions=cat("actions"), rewards=cat("rewards"),
               returns=cat("returns"), advantages=adv)
    if "agent_infos" in paths[0]:
        out["agent_infos"] = {k: np.concatenate([p["agent_infos"][k] for p in paths])
                              for k in paths[0]["agent_infos"]}
    return out, coeffs


def path_stats(all_paths):
    """Logging statistics of samplers/base.py:135-149."""
    # the reference uses Python sum() in the rewards' own dtype (float32 accumulates rounding);
    # the oracle sums in float64, parity tolerance on these logging stats is 1e-6 relative
    undisc = np.array([np.sum(np.asarray(p["rewards"], dtype=np.float64)) for p in all_paths])
    return dict(AverageDiscountedReturn=float(np.mean([p["returns"][0] for p in all_paths])),
                AverageReturn=float(np.mean(undisc)), NumTrajs=len(all_paths),
                StdReturn=float(np.std(undisc)), MaxReturn=float(np.max(undisc)),
                MinReturn=float(np.min(undisc)))


def process_samples_meta(paths_meta_batch, **kw):
    """All tasks + cross-task reward z-score.  Restates samplers/meta_sample_processor.py:8-49.

    Returns (list[M] of samples_data dicts with 'adj_avg_rewards', list[M] of coeffs, stats dict).
    """
    assert isinstance(paths_meta_batch, dict)
    out, coeffs_all, all_paths = [], [], []
    for _, paths in paths_meta_batch.items():
        sd, c = compute_samples_data(paths, **kw)
        out.append(sd)
        coeffs_all.append(c)
        all_paths.extend(paths)
    all_rew = np.concatenate([sd["rewards"] for sd in out])
    mu, sigma = np.mean(all_rew), np.std(all_rew)                                  # :40-41
    for sd in out:
        sd["adj_avg_rewards"] = (sd["rewards"] - mu) / (sigma + 1e-8)              # :43-44
    return out, coeffs_all, path_stats(all_paths)
