"""A stand-in for `optuna` (absent from this image, no network) - TEST INFRASTRUCTURE ONLY, kept in its own
directory so that nothing but tests/test_dropin_reference.py ever sees it.  It implements exactly what
run_examples/tune.py:136-225 touches: `create_study(direction, sampler)`, `samplers.TPESampler(seed)`,
`study.optimize(objective, n_trials)`, `trial.suggest_categorical / suggest_int / suggest_float`, and the
`best_trial / best_params / best_value` read-back.  Suggestions are the lower bound / first choice: the point of the
test is the CALL ORDER of a trial (fresh model + sampler + loader per fold, then `fit`, then `rank`), not the search."""


class _Trial:
    def __init__(self, number):
        self.number, self.params, self.value = number, {}, None

    def suggest_categorical(self, name, choices):
        self.params[name] = choices[0]
        return choices[0]

    def suggest_int(self, name, low, high, step=1):
        self.params[name] = int(low)
        return int(low)

    def suggest_float(self, name, low, high, step=None):
        self.params[name] = float(low)
        return float(low)


class _Study:
    def __init__(self, direction):
        self.sign = 1.0 if direction == "maximize" else -1.0
        self.trials = []

    def optimize(self, objective, n_trials):
        for k in range(int(n_trials)):
            t = _Trial(k)
            t.value = objective(t)
            self.trials.append(t)

    @property
    def best_trial(self):
        return max(self.trials, key=lambda t: self.sign * t.value)

    @property
    def best_params(self):
        return self.best_trial.params

    @property
    def best_value(self):
        return self.best_trial.value


class samplers:                                   # noqa: N801  (optuna.samplers.TPESampler)
    class TPESampler:
        def __init__(self, seed=None):
            self.seed = seed


def create_study(direction="minimize", sampler=None):
    return _Study(direction)
