"""PCA-assisted Bayesian optimisation on MI355X.

Same call surface as the reference's `PCA_BO`
(/root/reference/Algorithms/BayesianOptimization/PCA_BO.py:48-720): constructor keywords,
`__call__(problem, dim, bounds)`, result attributes, `TIME_PROFILES`, acquisition-name handling and
error behaviour.  The four places where the reference hands arithmetic to sklearn / botorch /
gpytorch / scipy are calls into libpcabo.so (HIP kernels for gfx950) here:

    reference                                              this file -> C ABI (include/pcabo.h)
    _calculate_weights + PCA().fit/transform (:316-408)    Context.wpca            pcabo_wpca
    SingleTaskGP(...) + lazy Gram/Cholesky (:502-545)      Context.gp_condition    pcabo_gp_condition
    optimize_acqf raw-sample scoring (:607)                Context.acq_eval        pcabo_acq_eval
    optimize_acqf multi-start L-BFGS-B (:607-614)          Context.optimize_acqf   pcabo_optimize_acqf
    pca.inverse_transform (:427)                           Context.inverse_map     pcabo_inverse_map

What stays on the host is exactly what the reference does in Python: ranking with numpy's argsort
(:330-333), the noise draw from numpy's global RNG (:376), Sobol / multinomial draws from torch's
global CPU generator (botorch initialisers), the out-of-bounds rule (:248-263) and bookkeeping.
There is no CPU fallback: importing this module needs libpcabo.so, running it needs a HIP device.
"""
from __future__ import annotations

import os
import warnings
from time import perf_counter
from typing import Callable, Optional, Union

import numpy as np

from pcabo import _native
from pcabo import initializers as _init
from pcabo import acqopt as _acqopt
from pcabo import gcguard as _gcguard
from .AbstractBayesianOptimizer import AbstractBayesianOptimizer

ALLOWED_ACQUISITION_FUNCTION_STRINGS = (
    "expected_improvement",
    "probability_of_improvement",
    "upper_confidence_bound",
)
ALLOWED_SHORTHAND_ACQUISITION_FUNCTION_STRINGS = {
    "EI": "expected_improvement",
    "PI": "probability_of_improvement",
    "UCB": "upper_confidence_bound",
}

# Model constants of the never-trained SingleTaskGP(MaternKernel(2.5)).
LENGTHSCALE = 0.6931471805599453     # softplus(0)
NOISE = 0.006737946999085467         # exp(-5), mode of the LogNormal(-4, 1) noise prior
OOB_PENALTY = 1000


class AnalyticAcquisitionFunction:
    """Descriptor of the acquisition the device kernels evaluate (stands in for botorch's class)."""
    acq_code = _native.ACQ_LOG_EI

    def __init__(self, model=None, best_f: float = 0.0, maximize: bool = True):
        self.model, self.best_f, self.maximize = model, best_f, maximize


class LogExpectedImprovement(AnalyticAcquisitionFunction):
    acq_code = _native.ACQ_LOG_EI


class ProbabilityOfImprovement(AnalyticAcquisitionFunction):
    acq_code = _native.ACQ_PI


class UpperConfidenceBound(AnalyticAcquisitionFunction):
    """The reference constructs its acquisition with `best_f=` (PCA_BO.py:199-203), which botorch's
    UpperConfidenceBound(model, beta, ...) does not accept: the first BO iteration raises TypeError.
    That behaviour is kept."""

    def __init__(self, model=None, beta=None, maximize: bool = True, **kwargs):
        if kwargs or beta is None:
            raise TypeError("UpperConfidenceBound.__init__() got an unexpected keyword argument 'best_f'")
        super().__init__(model, 0.0, maximize)


class _FittedPCA:
    """Read-only view of the device wPCA result with sklearn's attribute names (reference: `self.pca`)."""

    def __init__(self, components, mean, evr, k):
        self.components_ = components[:k]
        self.mean_ = mean
        self.explained_variance_ratio_ = evr
        self.n_components_ = components.shape[0]

    def transform(self, X):
        X = np.asarray(X, dtype=float)
        return X @ self.components_.T - self.mean_.reshape(1, -1) @ self.components_.T

    def inverse_transform(self, Z):
        return np.asarray(Z, dtype=float) @ self.components_ + self.mean_


class PCA_BO(AbstractBayesianOptimizer):
    TIME_PROFILES = ["SingleTaskGP", "optimize_acqf", "pca"]

    def __init__(self, budget: int, n_DoE: int = 0, n_components: int = 0, var_threshold: float = 0.95,
                 acquisition_function: str = "expected_improvement", random_seed: int = 43,
                 visualize: bool = False, **kwargs):
        self.__device = int(kwargs.pop("device", 0))
        self.__record_trace = bool(kwargs.pop("record_trace", False))
        # prefetch_noise: draw the NEXT iteration's 1e-8 noise matrix (numpy's global RNG, PCA_BO.py:376) on a helper
        # thread while the acquisition optimiser runs inside the library.  The numbers are the same, but they leave the
        # global stream BEFORE the objective of this iteration is evaluated - harmless exactly when the objective does
        # not draw from numpy's global RNG.  None = automatic: on for the in-repo BBOB problems, off otherwise.
        self.__prefetch = kwargs.pop("prefetch_noise", None)
        self.__prefetch_on = False
        self.__noise_thread, self.__noise_pending = None, False
        self.__noise_req = self.__noise_res = None
        # torch is only used for the Sobol / multinomial draws here; its default of one OpenMP worker per visible core
        # (hundreds on a GPU host) leaves spinning workers that starve the host threads driving the device loop
        # (measured: 300 us instead of 34 us per L-BFGS-B round).  Capped for the duration of a run; None = leave alone.
        self.__torch_threads = kwargs.pop("torch_threads", 4)
        self.__saved_torch_threads = None
        # gc_freeze: keep the interpreter's cyclic collector away from the loop (0.3-0.45 ms per iteration otherwise,
        # see pcabo/gcguard.py)
        # acq_kernel: "latency" (default: per-query kernels, resident across the evaluations of an optimize call - the
        # fastest for ONE run) or "group" (the throughput kernel the batched driver uses; a run then takes, bit for bit,
        # the path it takes inside a batch - pcabo.batchrun)
        self.__acq_kernel = str(kwargs.pop("acq_kernel", "latency"))
        if self.__acq_kernel not in ("latency", "group"):
            raise ValueError("acq_kernel must be 'latency' or 'group'")
        self.__gc_freeze = bool(kwargs.pop("gc_freeze", True))
        # resident=False: one launch per acquisition evaluation instead of the resident kernel of pcabo_optimize_acqf
        # (PCABO_OPT_RESIDENT; same arithmetic - the tests compare whole runs bit for bit)
        self.__resident = bool(kwargs.pop("resident", True))
        # fused_enqueue=False: pcabo_wpca and pcabo_gp_condition_begin as two calls instead of ONE enqueue of rows A-H (the
        # tests compare the two); early_scoring / engine_guess likewise switch the two host/device overlaps off
        self.__fused = bool(kwargs.pop("fused_enqueue", True))
        self.__early_scoring = bool(kwargs.pop("early_scoring", True))
        self.__speculate_engine = bool(kwargs.pop("engine_guess", True))
        self.__gc_entered = False
        super().__init__(budget, n_DoE, **kwargs)
        self.random_seed = random_seed
        smoke_test = os.environ.get("SMOKE_TEST")
        self.__torch_config = {
            "device": f"hip:{self.__device}",
            "dtype": np.float64,
            "SMOKE_TEST": smoke_test,
            "BATCH_SIZE": 3 if not smoke_test else 2,
            "NUM_RESTARTS": 10 if not smoke_test else 2,
            "RAW_SAMPLES": 512 if not smoke_test else 32,
        }
        self.__acq_func_class = None
        self.__acq_func = None
        self.acquisition_function_name = acquisition_function
        self.n_components = n_components
        self.var_threshold = var_threshold
        self.data_mean = None
        self.pca = None
        self.component_matrix = None
        self.explained_variance_ratio = None
        self.reduced_space_dim_num = None
        self.visualize = visualize
        if visualize:
            warnings.warn("visualize=True: the GIF visualiser of the reference is not part of the MI355X path; ignored.")
        self.__z_evals = []
        self.__gp_pending = False
        self.__engine_ready = None
        self.__X_buf, self.__X_rows = None, 0
        self.__ctx: Optional[_native.Context] = None
        self.lbfgsb_info = []          # per iteration: (iterations, evaluations, warnflag, task) per restart group
        self.trace = []                # record_trace=True: per iteration RNG states, restart candidates/values
        self.phase_breakdown = {"sobol": 0.0, "raw_eval": 0.0, "init_pick": 0.0, "lbfgsb": 0.0}   # inside optimize_acqf

    def __str__(self):
        return "This is an instance of a PCA-assisted BO Optimizer"

    # ---------------------------------------------------------------------------------------------
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

    # The three pieces of `__call__`, exposed so that a driver (bench.py, the sharded runner) can time or
    # interleave single BO iterations; together they are exactly the reference's loop (PCA_BO.py:140-310).
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
        if self.__acq_kernel == "group":
            self.__ctx.set_option(_native.OPT_GROUP_ACQ, 1)
        if not self.__resident:
            self.__ctx.set_option(_native.OPT_RESIDENT, 0)
        self.__X_buf, self.__X_rows = None, 0
        if self.__prefetch is None:
            from pcabo.bbob import BBOBProblem
            self.__prefetch_on = isinstance(getattr(problem, "_problem", problem), BBOBProblem)
        else:
            self.__prefetch_on = bool(self.__prefetch)
        if self.__record_trace:            # the recorded per-iteration RNG state must be the one BEFORE the draw
            self.__prefetch_on = False

    def _bo_iteration(self, problem, **kwargs) -> None:
        if self.__record_trace:
            import torch
            self.trace.append({"n": len(self.f_evals), "numpy_state": np.random.get_state(),
                               "torch_state": torch.get_rng_state(), "best_f": self.current_best})
        self._transform_points_to_reduced_space()
        self._initialize_model(**kwargs)
        self.__ctx.match_best_f_dtype(self.current_best)      # float32 like torch.as_tensor(python float), or all 64 bits
        self.acquisition_function = self.acquisition_function_class(
            model=self.__ctx, best_f=self.current_best, maximize=self.maximization)
        new_z = self.optimize_acqf_and_get_observation()
        for new_z_arr in new_z:
            if self.number_of_function_evaluations >= self.budget:
                break
            new_x = self._transform_point_to_original_space(np.asarray(new_z_arr).ravel())
            outside = not np.all(new_x >= self.bounds[:, 0]) or not np.all(new_x <= self.bounds[:, 1])
            if outside and self.verbose:
                print(f"Warning: PCA transformed point {new_x} was out of bounds, clipping to boundary")
            self.x_evals.append(new_x)
            self.__z_evals.append(np.asarray(new_z_arr).ravel())
            # out-of-box candidates are not evaluated; they cost budget and a fixed penalty (PCA_BO.py:260-263)
            new_f = (-OOB_PENALTY if self.maximization else OOB_PENALTY) if outside else problem(new_x)
            if self._pbar is not None:
                self._pbar.update(1)
            self.f_evals.append(new_f)
            self.number_of_function_evaluations += 1
            if self.verbose and ((self.maximization and new_f > self.current_best) or
                                 (not self.maximization and new_f < self.current_best)):
                print(f"Found better solution: {new_f}")
                print(f"At point: {new_x}")
        self.assign_new_best()
        if self.verbose:
            print(f"Evaluations: {self.number_of_function_evaluations}/{self.budget}",
                  f"Best: x:{self.x_evals[self.current_best_index]} y:{self.current_best}", flush=True)

    def _finish(self) -> None:
        self._stop_noise_worker()
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
            print("Optimization Process finalized!")
        self.restore_random_states()

    @property
    def device_context(self):
        """The live libpcabo context of a run in progress (profiling hooks); None outside a run."""
        return self.__ctx

    def assign_new_best(self):
        super().assign_new_best()

    # ---- row A (host part): ranks exactly as numpy gives them ------------------------------------
    def _calculate_ranks(self) -> np.ndarray:
        f = np.array(self.f_evals)
        return np.argsort(np.argsort(-f if self.maximization else f)) + 1

    def _calculate_weights(self) -> np.ndarray:
        pre = np.log(len(self.f_evals)) - np.log(self._calculate_ranks())
        return pre / pre.sum()

    # ---- rows A-C ---------------------------------------------------------------------------------
    def _design_matrix(self) -> np.ndarray:
        """x_evals as one contiguous n x d array; rows already copied stay (the list only grows during a run), which
        avoids re-stacking several hundred small arrays every iteration."""
        n = len(self.x_evals)
        buf = self.__X_buf
        if buf is None or buf.shape[0] < n or self.__X_rows > n:
            buf = np.empty((max(n, self.budget), self.dimension), dtype=np.float64)
            self.__X_buf, self.__X_rows = buf, 0
        for i in range(self.__X_rows, n):
            buf[i] = self.x_evals[i]
        self.__X_rows = n
        return buf[:n]

    def _take_noise(self, shape) -> np.ndarray:
        if not self.__noise_pending:
            return np.random.normal(0, 1e-8, size=shape)
        nz = self.__noise_res.get()
        self.__noise_pending = False
        if isinstance(nz, BaseException):
            raise nz
        if nz.shape != tuple(shape):
            raise RuntimeError("noise prefetch out of step with the run (a draw of another shape left the RNG)")
        return nz

    def _noise_worker(self, req, res) -> None:
        while True:
            shape = req.get()
            if shape is None:
                return
            try:
                res.put(np.random.normal(0, 1e-8, size=shape))   # numpy releases the GIL while it generates
            except BaseException as e:  # noqa: BLE001 - handed to the thread that asked
                res.put(e)

    def _prefetch_noise(self) -> None:
        """Called once the current iteration's noise is consumed: the next iteration (if there is one) has one more
        point.  One worker thread per run (creating a thread per iteration cost 0.24 ms of every iteration)."""
        n = len(self.x_evals)
        if not self.__prefetch_on or self.__noise_pending or n + 1 >= self.budget:
            return
        if self.__noise_thread is None:
            import queue
            import threading
            self.__noise_req, self.__noise_res = queue.SimpleQueue(), queue.SimpleQueue()
            self.__noise_thread = threading.Thread(target=self._noise_worker, args=(self.__noise_req, self.__noise_res),
                                                   daemon=True)
            self.__noise_thread.start()
        self.__noise_pending = True
        self.__noise_req.put((n + 1, self.dimension))

    def _stop_noise_worker(self) -> None:
        if self.__noise_thread is None:
            return
        self.__noise_req.put(None)
        self.__noise_thread.join()
        self.__noise_thread, self.__noise_pending = None, False
        self.__noise_req = self.__noise_res = None

    def _transform_points_to_reduced_space(self) -> None:
        if len(self.x_evals) < 2:
            if len(self.x_evals) == 1:
                self.__z_evals = [np.zeros(1)]
            return
        X = self._design_matrix()
        ranks = self._calculate_ranks()
        noise = self._take_noise(X.shape)                      # same draw, same global RNG as the reference
        start = perf_counter()
        if self.__fused:
            # rows A-H in one enqueue: the GP conditioning is queued right behind the projection and runs while the
            # wPCA results travel back (`_initialize_model` then has nothing left to launch)
            self.__ctx.wpca_gp_condition(
                X, np.array(self.f_evals, dtype=np.float64), ranks=ranks, maximize=self.maximization,
                var_threshold=self.var_threshold, n_components=self.n_components, noise=noise,
                lengthscale=LENGTHSCALE, gp_noise=NOISE, kernel=_native.KERNEL_MATERN52, collect=False)
            self.__gp_pending = True
            # While the device runs the eigen-decomposition the host builds the scrambled Sobol engine of this
            # iteration's initial-condition draw (0.15 ms) - with LAST iteration's k, which is this iteration's k
            # almost always.  Same draws from torch's global generator at the same place of its stream (nothing else
            # consumes it between here and the optimiser); if k did change the generator is put back and the engine
            # is built later, as without the guess.
            self.__engine_ready, guess, saved = None, None, None
            k_prev = self.reduced_space_dim_num
            if self.__speculate_engine and k_prev:
                import torch
                saved = torch.get_rng_state()
                guess = _init.scrambled_sobol_engine(int(k_prev))
            res = self.__ctx.wpca_results()
            if guess is not None:
                if res["k"] == k_prev:
                    self.__engine_ready = guess
                else:
                    torch.set_rng_state(saved)
        else:
            res = self.__ctx.wpca(X, ranks=ranks, maximize=self.maximization, var_threshold=self.var_threshold,
                                  n_components=self.n_components, noise=noise, want_Z=False, want_full=True)
        self.timing_logs["pca"].append(perf_counter() - start)
        self.data_mean = res["data_mean"]
        self.component_matrix = res["components"]
        self.explained_variance_ratio = res["evr"]
        self.reduced_space_dim_num = res["k"]
        self.pca = _FittedPCA(res["components"], res["pca_mean"], res["evr"], res["k"])
        if self.verbose:
            k = res["k"]
            print(f"Using {k} principal components with {np.sum(res['evr'][:k]) * 100:.2f}% explained variance")
        # the reduced coordinates stay on the device (the reference keeps them in `__z_evals` only to
        # rebuild bounds and the GP input, both of which happen on the device here)
        self.__z_evals = [res["k"]] * X.shape[0]

    # ---- rows D-H ---------------------------------------------------------------------------------
    def _initialize_model(self, **kwargs):
        if not self.__z_evals:
            return
        start = perf_counter()
        if self.__gp_pending:              # already enqueued together with the wPCA
            self.timing_logs["SingleTaskGP"].append(perf_counter() - start)
            return
        # enqueue only: the device conditions the GP while the host prepares the Sobol engine (gp_wait below)
        self.__ctx.gp_condition(np.array(self.f_evals, dtype=np.float64), lengthscale=LENGTHSCALE, noise=NOISE,
                                kernel=_native.KERNEL_MATERN52, wait=False)
        self.__gp_pending = True
        self.timing_logs["SingleTaskGP"].append(perf_counter() - start)

    # ---- rows J-N ---------------------------------------------------------------------------------
    def optimize_acqf_and_get_observation(self) -> np.ndarray:
        ctx, cfg = self.__ctx, self.__torch_config
        acq = self.acquisition_function
        num_restarts, raw_samples, batch_limit = cfg["NUM_RESTARTS"], cfg["RAW_SAMPLES"], 5
        start = perf_counter()
        # Like the reference, where gpytorch's Gram/Cholesky happen lazily inside optimize_acqf, the wait for the
        # conditioning is accounted here; the scrambled Sobol engine (needs only k) is built meanwhile.
        engine, self.__engine_ready = self.__engine_ready, None
        if engine is None:
            engine = _init.scrambled_sobol_engine(ctx.k)
        bounds = ctx.acq_bounds()          # needs only the statistics kernel, not the factorisation
        t0 = perf_counter()
        raw = _init.draw_sobol(bounds, raw_samples, engine)
        self.phase_breakdown["sobol"] = self.phase_breakdown.get("sobol", 0.0) + perf_counter() - t0
        raw_vals = None
        if self.__gp_pending and self.__early_scoring:
            # the raw samples are scored right behind the conditioning on the stream: one wait for both
            t0 = perf_counter()
            raw_vals = ctx.gp_wait_eval(raw, acq.best_f, acq.maximize, acq.acq_code)
            self.__gp_pending = False
            self.phase_breakdown["raw_eval"] = self.phase_breakdown.get("raw_eval", 0.0) + perf_counter() - t0
        elif self.__gp_pending:
            ctx.gp_wait()
            self.__gp_pending = False
        new_z, cand, vals, info = _acqopt.optimize_acqf(
            ctx, bounds, acq.best_f, acq.maximize, acq.acq_code, num_restarts, raw_samples, batch_limit, 200,
            raw=raw, raw_vals=raw_vals, breakdown=self.phase_breakdown,
            trace=self.trace[-1] if self.__record_trace else None,
            before_lbfgsb=self._prefetch_noise)   # the draw overlaps with the optimiser's time inside the library; started
        # any earlier it shares the core with the torch ops of the initial pick and doubles their time (0.10 -> 0.21 ms)
        self.timing_logs["optimize_acqf"].append(perf_counter() - start)
        self.lbfgsb_info.append(info)
        return new_z

    # ---- row O ------------------------------------------------------------------------------------
    def _transform_point_to_original_space(self, z: np.ndarray) -> np.ndarray:
        if self.pca is None:
            return np.random.uniform(self.bounds[:, 0], self.bounds[:, 1])
        return self.__ctx.inverse_map(z)

    def __repr__(self):
        return super().__repr__()

    def reset(self):
        super().reset()
        self.__z_evals = []
        self.pca = None
        self.explained_variance_ratio = None

    # ---- acquisition-name plumbing (reference :643-720) -------------------------------------------
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
            raise ValueError("Oddly defined name")
        self.set_acquisition_function_subclass()

    def set_acquisition_function_subclass(self) -> None:
        name = self.__acquisition_function_name
        if name == ALLOWED_ACQUISITION_FUNCTION_STRINGS[0]:
            self.__acq_func_class = LogExpectedImprovement
        elif name == ALLOWED_ACQUISITION_FUNCTION_STRINGS[1]:
            self.__acq_func_class = ProbabilityOfImprovement
        elif name == ALLOWED_ACQUISITION_FUNCTION_STRINGS[2]:
            self.__acq_func_class = UpperConfidenceBound

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
            raise AttributeError("Acquisition function does not inherit from 'AnalyticAcquisitionFunction'",
                                 name="acquisition_function", obj=self.__acq_func)
