"""Plain (full-dimensional) Bayesian optimisation on MI355X.

Same call surface as the reference's `Vanilla_BO`
(/root/reference/Algorithms/BayesianOptimization/Vanilla_BO.py:39-301).  It is the PCA_BO loop without the PCA:
the exact GP lives on the raw d-dimensional points (the reference disables `Normalize` there,
Vanilla_BO.py:188-194, so the kernel sees raw coordinates), the acquisition is optimised inside the problem's
box (Vanilla_BO.py:206-213) and there is no out-of-bounds rule.  It runs on the same libpcabo kernels:

    SingleTaskGP(...) + lazy Gram/Cholesky (:166-196, :206)   Context.gp_condition(Z = X, identity Normalize bounds)
    optimize_acqf (:206-213)                                  pcabo.acqopt.optimize_acqf (device scoring + L-BFGS-B)
"""
from __future__ import annotations

import os
from time import perf_counter
from typing import Callable, Optional, Union

import numpy as np

from pcabo import _native
from pcabo import acqopt as _acqopt
from pcabo import gcguard as _gcguard
from pcabo import initializers as _init
from .AbstractBayesianOptimizer import AbstractBayesianOptimizer
from .PCA_BO import (ALLOWED_ACQUISITION_FUNCTION_STRINGS, ALLOWED_SHORTHAND_ACQUISITION_FUNCTION_STRINGS,
                     AnalyticAcquisitionFunction, LogExpectedImprovement, ProbabilityOfImprovement,
                     UpperConfidenceBound, LENGTHSCALE, NOISE)


class Vanilla_BO(AbstractBayesianOptimizer):
    TIME_PROFILES = ["SingleTaskGP", "optimize_acqf"]

    def __init__(self, budget: int, n_DoE: int = 0, acquisition_function: str = "expected_improvement",
                 random_seed: int = 43, **kwargs):
        self.__device = int(kwargs.pop("device", 0))
        self.__record_trace = bool(kwargs.pop("record_trace", False))
        self.__torch_threads = kwargs.pop("torch_threads", 4)      # see PCA_BO: spinning OpenMP workers starve the loop
        self.__saved_torch_threads = None
        self.__gc_freeze = bool(kwargs.pop("gc_freeze", True))      # see pcabo/gcguard.py
        self.__gc_entered = False
        self.__resident = bool(kwargs.pop("resident", True))         # see PCA_BO: False = one launch per evaluation
        super().__init__(budget, n_DoE, **kwargs)
        self.random_seed = random_seed
        smoke_test = os.environ.get("SMOKE_TEST")
        self.__torch_config = {
            "device": f"hip:{self.__device}", "dtype": np.float64, "SMOKE_TEST": smoke_test,
            "BATCH_SIZE": 3 if not smoke_test else 2,
            "NUM_RESTARTS": 10 if not smoke_test else 2,
            "RAW_SAMPLES": 512 if not smoke_test else 32,
        }
        self.__acq_func_class = None
        self.__acq_func = None
        self.acquisition_function_name = acquisition_function
        self.__ctx: Optional[_native.Context] = None
        self.lbfgsb_info = []
        self.trace = []
        self.phase_breakdown = {}

    def __str__(self):
        return "This is an instance of a Vanilla BO Optimizer"

    def __call__(self, problem: Union[Callable, object], dim: Optional[int] = -1,
                 bounds: Optional[np.ndarray] = None, **kwargs) -> None:
        try:
            self._start(problem, dim, bounds, **kwargs)      # inside the try: whatever it switched on is switched off again
            for _ in range(self.budget - self.n_DoE):
                if self.number_of_function_evaluations >= self.budget:
                    break
                self._bo_iteration(problem, **kwargs)
        finally:
            self._finish()

    def _start(self, problem, dim=-1, bounds=None, **kwargs) -> None:
        if self.__torch_threads is not None:
            import torch
            self.__saved_torch_threads = torch.get_num_threads()
            if self.__saved_torch_threads > int(self.__torch_threads):
                torch.set_num_threads(int(self.__torch_threads))
        if self.__gc_freeze and not self.__gc_entered:
            _gcguard.enter()
            self.__gc_entered = True
        self.impose_random_seed()
        AbstractBayesianOptimizer.__call__(self, problem, dim, bounds, **kwargs)
        if self._pbar is not None:
            self._pbar.update(self.n_DoE)
        self.__ctx = _native.Context(max_n=self.budget, max_d=self.dimension,
                                     max_q=max(self.__torch_config["RAW_SAMPLES"], 16), device=self.__device)
        if not self.__resident:
            self.__ctx.set_option(_native.OPT_RESIDENT, 0)
        d = self.dimension
        self.__identity = np.vstack([np.zeros(d), np.ones(d)])          # Normalize is switched off in the reference
        self.__box = np.ascontiguousarray(self.bounds.T, dtype=np.float64)   # 2 x d search box

    def _bo_iteration(self, problem, **kwargs) -> None:
        if self.__record_trace:
            import torch
            self.trace.append({"n": len(self.f_evals), "numpy_state": np.random.get_state(),
                               "torch_state": torch.get_rng_state(), "best_f": self.current_best})
        self._initialise_model(**kwargs)
        self.__ctx.match_best_f_dtype(self.current_best)      # float32 like torch.as_tensor(python float), or all 64 bits
        self.acquisition_function = self.acquisition_function_class(
            model=self.__ctx, best_f=self.current_best, maximize=self.maximization)
        new_x = self.optimize_acqf_and_get_observation()
        for new_x_arr in new_x:
            if self.number_of_function_evaluations >= self.budget:
                break
            x = np.asarray(new_x_arr, dtype=np.float64).ravel()
            self.x_evals.append(x)
            new_f = problem(x)
            if self._pbar is not None:
                self._pbar.update(1)
            self.f_evals.append(new_f)
            self.number_of_function_evaluations += 1
        self.assign_new_best()
        if self.verbose:
            print(f"Evaluations: {self.number_of_function_evaluations}/{self.budget}",
                  f"Best: x:{self.x_evals[self.current_best_index]} y:{self.current_best}", flush=True)

    def _finish(self) -> None:
        if self.__gc_entered:
            _gcguard.leave()
            self.__gc_entered = False
        if self.__saved_torch_threads is not None:
            import torch
            torch.set_num_threads(self.__saved_torch_threads)
            self.__saved_torch_threads = None
        if self.__ctx is not None:
            self.__ctx.close()
            self.__ctx = None
        if self.verbose:
            print("Optimisation Process finalized!")
        self.restore_random_states()

    def assign_new_best(self):
        super().assign_new_best()

    def _initialise_model(self, **kwargs):
        X = np.array(self.x_evals, dtype=np.float64).reshape((-1, self.dimension))
        y = np.array(self.f_evals, dtype=np.float64)
        start = perf_counter()
        self.__ctx.gp_condition(y, Z=X, norm_bounds=self.__identity, lengthscale=LENGTHSCALE, noise=NOISE,
                                kernel=_native.KERNEL_MATERN52, wait=False)
        self.timing_logs["SingleTaskGP"].append(perf_counter() - start)

    def optimize_acqf_and_get_observation(self) -> np.ndarray:
        ctx, cfg, acq = self.__ctx, self.__torch_config, self.acquisition_function
        start = perf_counter()
        engine = _init.scrambled_sobol_engine(self.dimension)      # built and drawn while the device conditions the GP
        raw = _init.draw_sobol(self.__box, cfg["RAW_SAMPLES"], engine)
        t0 = perf_counter()                # the raw samples are scored right behind the conditioning: one wait for both
        raw_vals = ctx.gp_wait_eval(raw, acq.best_f, acq.maximize, acq.acq_code)
        self.phase_breakdown["raw_eval"] = self.phase_breakdown.get("raw_eval", 0.0) + perf_counter() - t0
        new_x, cand, vals, info = _acqopt.optimize_acqf(
            ctx, self.__box, acq.best_f, acq.maximize, acq.acq_code, cfg["NUM_RESTARTS"], cfg["RAW_SAMPLES"], 5, 200,
            raw=raw, raw_vals=raw_vals, breakdown=self.phase_breakdown,
            trace=self.trace[-1] if self.__record_trace else None)
        self.timing_logs["optimize_acqf"].append(perf_counter() - start)
        self.lbfgsb_info.append(info)
        return new_x

    def __repr__(self):
        return super().__repr__()

    def reset(self):
        super().reset()

    @property
    def device_context(self):
        return self.__ctx

    @property
    def torch_config(self) -> dict:
        return self.__torch_config

    @property
    def acquisition_function_name(self) -> str:
        return self.__acquisition_function_name

    @acquisition_function_name.setter
    def acquisition_function_name(self, new_name: str) -> None:
        new_name = new_name.strip()
        if new_name in ALLOWED_SHORTHAND_ACQUISITION_FUNCTION_STRINGS:
            self.__acquisition_function_name = ALLOWED_SHORTHAND_ACQUISITION_FUNCTION_STRINGS[new_name]
        elif new_name.lower() in ALLOWED_ACQUISITION_FUNCTION_STRINGS:
            self.__acquisition_function_name = new_name
        else:
            raise ValueError(f"Oddly defined name {new_name}")
        self.set_acquisition_function_subclass()

    def set_acquisition_function_subclass(self) -> None:
        name = self.__acquisition_function_name
        self.__acq_func_class = {ALLOWED_ACQUISITION_FUNCTION_STRINGS[0]: LogExpectedImprovement,
                                 ALLOWED_ACQUISITION_FUNCTION_STRINGS[1]: ProbabilityOfImprovement,
                                 ALLOWED_ACQUISITION_FUNCTION_STRINGS[2]: UpperConfidenceBound}[name]

    @property
    def acquisition_function_class(self) -> Callable:
        return self.__acq_func_class

    @property
    def acquisition_function(self) -> AnalyticAcquisitionFunction:
        return self.__acq_func

    @acquisition_function.setter
    def acquisition_function(self, new_acquisition_function: AnalyticAcquisitionFunction) -> None:
        if issubclass(type(new_acquisition_function), AnalyticAcquisitionFunction):
            self.__acq_func = new_acquisition_function
        else:
            raise AttributeError("Cannot assign the acquisition function as this does not inherit from the class "
                                 "`AnalyticAcquisitionFunction`", name="acquisition_function", obj=self.__acq_func)
