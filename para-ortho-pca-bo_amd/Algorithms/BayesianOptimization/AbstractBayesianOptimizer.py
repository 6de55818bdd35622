"""Initial design + evaluation bookkeeping shared by the Bayesian optimisers.

Host-side mirror of /root/reference/Algorithms/BayesianOptimization/AbstractBayesianOptimizer.py
(:8-103 `LHS_sampler`, :106-270 `AbstractBayesianOptimizer`).  The Latin-hypercube draw is pyDOE's
`lhs` restated in `pcabo.lhs` (pyDOE is not installed here): "center" pinned by the reference's committed runs, the other
three criteria restated from the published algorithm (parity unpinned).
"""
from __future__ import annotations

from abc import abstractmethod
from typing import List, Optional

import numpy as np

from ..AbstractAlgorithm import AbstractAlgorithm
from pcabo.lhs import lhs

_CRITERIA = ("center", "maximin", "centermaximin", "correlation")
_SHORT = {"c": "center", "m": "maximin", "cm": "centermaximin", "corr": "correlation"}


class LHS_sampler:
    def __init__(self, criterion: str = "correlation", iterations: int = 1000, sample_zero: bool = False):
        self.criterion = criterion
        self.iterations = iterations
        self.sample_zero = sample_zero

    def __call__(self, dim: int, n_samples: int) -> np.ndarray:
        points = lhs(dim, n_samples, criterion=self.criterion, iterations=self.iterations)
        if self.sample_zero:
            points[0, :] = 0.0
        return points.reshape((n_samples, dim))

    @property
    def criterion(self) -> str:
        return self.__criterion

    @criterion.setter
    def criterion(self, value: str) -> None:
        if not isinstance(value, str):
            raise ValueError("The new criterion is not a string!")
        value = value.lower().strip()
        value = _SHORT.get(value, value)
        if value not in _CRITERIA:
            raise ValueError("The criterion is not matching the set ones!")
        self.__criterion = value

    @property
    def iterations(self) -> int:
        return self.__iterations

    @iterations.setter
    def iterations(self, value: int) -> None:
        if value > 0:
            self.__iterations = int(value)
        else:
            raise ValueError("Negative iterations not allowed")

    @property
    def sample_zero(self) -> bool:
        return self.__sample_zero

    @sample_zero.setter
    def sample_zero(self, value: bool) -> None:
        self.__sample_zero = value


class AbstractBayesianOptimizer(AbstractAlgorithm):
    def __init__(self, budget: int, n_DoE: Optional[int] = 0, **kwargs):
        super().__init__(**kwargs)
        self.budget = budget
        self.n_DoE = n_DoE
        params = {"criterion": "center", "iterations": 1000, "sample_zero": False}
        for key, item in kwargs.items():
            if key.lower().strip() == "doe_parameters" and isinstance(item, dict):
                params.update(item)
        self.__lhs_sampler = LHS_sampler(params["criterion"], params["iterations"], params["sample_zero"])
        self.__x_evals: List[np.ndarray] = []
        self.__f_evals: List[float] = []

    def __str__(self):
        pass

    def __call__(self, problem, dim: int, bounds: np.ndarray, **kwargs) -> None:
        """Draw and evaluate the initial design (reference :142-176)."""
        super().__call__(problem, dim, bounds, **kwargs)
        if not isinstance(self.n_DoE, int) or self.n_DoE == 0:
            self.n_DoE = self.dimension
        unit = self.lhs_sampler(self.dimension, self.n_DoE)
        span = self.bounds[:, 1] - self.bounds[:, 0]
        for point in span * unit + self.bounds[:, 0]:
            self.__x_evals.append(point)
            self.__f_evals.append(problem(point))
        self.assign_new_best()
        self.number_of_function_evaluations = self.n_DoE
        if self.verbose:
            print("After Initial sampling...",
                  f"Current Best: x:{self.__x_evals[self.current_best_index]} y:{self.current_best}", flush=True)

    @abstractmethod
    def assign_new_best(self):
        self.current_best = max(self.__f_evals) if self.maximization else min(self.__f_evals)
        self.current_best_index = self.__f_evals.index(self.current_best, self.current_best_index)

    def __repr__(self):
        return object.__repr__(self)

    def reset(self) -> None:
        super().reset()
        self.__x_evals = []
        self.__f_evals = []

    @property
    def budget(self) -> int:
        return self.__budget

    @budget.setter
    def budget(self, value: int) -> None:
        assert value > 0
        self.__budget = int(value)

    @property
    def n_DoE(self):
        return self.__n_DoE

    @n_DoE.setter
    def n_DoE(self, value) -> None:
        self.__n_DoE = int(value) if value >= 0 else None

    @property
    def lhs_sampler(self) -> LHS_sampler:
        return self.__lhs_sampler

    @property
    def f_evals(self) -> List[float]:
        return self.__f_evals

    @property
    def x_evals(self) -> List[np.ndarray]:
        return self.__x_evals
