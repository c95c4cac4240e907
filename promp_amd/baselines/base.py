class Baseline:
    """Reward baseline interface (reference: meta_policy_search/baselines/base.py:4-52)."""

    def get_param_values(self):
        raise NotImplementedError

    def set_params(self, value):
        raise NotImplementedError

    def fit(self, paths):
        raise NotImplementedError

    def predict(self, path):
        raise NotImplementedError

    def log_diagnostics(self, paths, prefix):
        pass
