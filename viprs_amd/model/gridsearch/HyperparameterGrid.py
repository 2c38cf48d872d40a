"""Hyper-parameter grids for the spike-and-slab model, with the constructor and the `generate_*`
/ `combine_grids` / `to_table` surface of viprs/model/gridsearch/HyperparameterGrid.py.

Grid definitions (reference lines): heritability percentiles of N(h2_est, h2_se) between the 10th and
90th (:110-150); sigma_epsilon = 1 - h2 (:152-166); tau_beta = 0.01 n_snps / h2 (:168-182);
pi log-spaced in [max(10 / n_snps, 1e-5), min(1e4 / n_snps, max_pi)] (:184-208); lambda_min =
{0} U logspace(-4, 1) (:210-227).  Parameters are combined with itertools.product in the order they
were added, so the first one varies slowest (:229-245).
"""
import itertools

import numpy as np
import pandas as pd

_ORDER = ("sigma_epsilon", "tau_beta", "pi", "lambda_min")


class HyperparameterGrid:

    def __init__(self, sigma_epsilon_grid=None, sigma_epsilon_steps=None, tau_beta_grid=None, tau_beta_steps=None,
                 pi_grid=None, pi_steps=None, lambda_min_grid=None, lambda_min_steps=None, h2_est=None, h2_se=None,
                 n_snps=1e6):
        self.h2_est = h2_est or 0.1
        self.h2_se = h2_se or 0.1
        self.n_snps = n_snps
        self._search_params = []
        given = dict(sigma_epsilon=(sigma_epsilon_grid, sigma_epsilon_steps), tau_beta=(tau_beta_grid, tau_beta_steps),
                     pi=(pi_grid, pi_steps), lambda_min=(lambda_min_grid, lambda_min_steps))
        for name in _ORDER:
            grid, steps = given[name]
            setattr(self, name, grid)
            if grid is not None:
                self._search_params.append(name)
            elif steps is not None:
                getattr(self, f"generate_{name}_grid")(steps=steps)

    def _register(self, name, values):
        setattr(self, name, values)
        if name not in self._search_params:
            self._search_params.append(name)

    def _generate_h2_grid(self, steps=5):
        assert steps > 0 and self.h2_est is not None
        se = self.h2_est * 0.5 if self.h2_se is None else self.h2_se
        assert 0.0 < self.h2_est < 1.0 and se > 0
        from scipy.stats import norm
        lo = max(0.1, norm.cdf(1e-5, loc=self.h2_est, scale=se))
        hi = min(0.9, norm.cdf(1.0 - 1e-5, loc=self.h2_est, scale=se))
        return norm.ppf(np.linspace(lo, hi, steps), loc=self.h2_est, scale=se)

    def generate_sigma_epsilon_grid(self, steps=5):
        assert steps > 0
        self._register("sigma_epsilon", 1.0 - self._generate_h2_grid(steps))

    def generate_tau_beta_grid(self, steps=5):
        assert steps > 0
        self._register("tau_beta", 0.01 * self.n_snps / self._generate_h2_grid(steps))

    def generate_pi_grid(self, steps=5, max_pi=0.2):
        assert steps > 0
        lo = np.log10(max(10.0 / self.n_snps, 1e-5))
        hi = np.log10(min(10000 / self.n_snps, max_pi))
        assert lo < hi
        self._register("pi", np.logspace(lo, hi, steps))

    def generate_lambda_min_grid(self, steps=5, emp_lambda_min=None):
        assert steps > 0
        grid = np.concatenate([[0.0], np.logspace(-4, 1.0, steps - 1)])
        if emp_lambda_min is not None:
            grid = grid * emp_lambda_min
        self._register("lambda_min", grid)

    def combine_grids(self):
        names = [n for n in _ORDER if n in self._search_params and getattr(self, n) is not None]
        if not names:
            raise ValueError("All the grids are empty!")
        return [dict(zip(names, combo)) for combo in itertools.product(*[getattr(self, n) for n in names])]

    def to_table(self):
        return pd.DataFrame(self.combine_grids())
