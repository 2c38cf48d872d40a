"""Optimisation bookkeeping with the attribute surface of the reference's
viprs/utils/OptimizeResult.py (``optim_result.nit / success / message / stop_iteration / fun``)."""


class ConditionStreak:
    """Counts consecutive iterations on which a condition held (OptimizeResult.py:5-40)."""

    def __init__(self):
        self.counter = 0
        self._last = 0

    def update(self, condition, iteration):
        self.counter = self.counter + 1 if (condition and iteration == self._last + 1) else 0
        self._last = iteration


class OptimizeResult:
    def __init__(self):
        self.reset()

    def reset(self):
        self.message = None
        self.stop_iteration = False
        self.success = False
        self.fun = None
        self.nit = 0
        self.error_on_termination = False
        self._drop_at = None
        self.oscillation_counter = 0

    # aliases used by callers of the reference class
    @property
    def iterations(self):
        return self.nit

    @property
    def objective(self):
        return self.fun

    @property
    def converged(self):
        return self.success

    @property
    def valid_optim_result(self):
        return self.success or (self.stop_iteration and not self.error_on_termination)

    def _reset_oscillation_counter(self):
        self.oscillation_counter = 0

    def update(self, fun, stop_iteration=False, success=False, message=None, increment=True):
        # drops of the objective on consecutive iterations count as oscillations (:118-129)
        if self.fun is not None and fun < self.fun:
            if self._drop_at is not None and self.nit - self._drop_at == 1:
                self.oscillation_counter += 1
            self._drop_at = self.nit + 1
        elif self._drop_at is not None and self.nit > self._drop_at:
            self.oscillation_counter = 0
        self.fun, self.stop_iteration, self.success, self.message = fun, stop_iteration, success, message
        self.nit += int(increment)
        if stop_iteration and not success and "Maximum iterations" not in (message or ""):
            self.error_on_termination = True

    def __str__(self):
        return str(self.__dict__)
