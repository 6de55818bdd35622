"""B independent PCA_BO runs advancing in lock-step on one GPU (the batched multi-run driver).

The reference's outer loop is a list of independent runs - 30 instances per (function, dimension) cell, each with its
own optimiser object and seed (/root/reference/Algorithms/Experiment/ExperimentRunner.py:137-183).  One run alone is
latency-bound on a GPU: its matrices are a few MB and every phase is a short dependency chain.  Here the runs of a
cell (same dimension, budget and DoE size, hence the same n at every iteration) advance TOGETHER through
`pcabo._native.Batch`: one launch sequence conditions the GPs of all runs (blockIdx.z = run), one launch scores all raw
samples, and the L-BFGS-B rounds of all runs share acquisition launches.

Every run is the reference's run: the loop below is `PCA_BO.__call__` (PCA_BO.py:140-310) per run, with the numpy
and torch GLOBAL generators of the reference replaced by one `RandomState` / `torch.Generator` per run seeded like the
reference seeds its globals (AbstractAlgorithm.py:310-328) - the same streams, so a run here takes, bit for bit, the
path the same run takes alone in `Algorithms.PCA_BO` (tests/test_gpu_batch.py).
"""
from __future__ import annotations

import warnings
from time import perf_counter
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _native
from .hostrng import HostMT
from . import initializers as _init
from .lhs import lhs_center

LENGTHSCALE = 0.6931471805599453     # softplus(0)
NOISE = 0.006737946999085467         # exp(-5)
OOB_PENALTY = 1000


class _Share:
    """One run's part of a pool job that returns {run: value} (looks like the future of that part)."""
    __slots__ = ("job", "key")

    def __init__(self, job, key):
        self.job, self.key = job, key

    def result(self):
        return self.job.result()[self.key]


def device_mode_covers(dimension: int, budget: int) -> bool:
    """Can every iteration of a run of this shape go through the device-resident optimiser (acq_kernel="device")?  n never exceeds
    the budget and k never exceeds the dimension; pcabo_device_lbfgsb_limits gives the kernel's bounds."""
    max_n, max_k, _ = _native.device_lbfgsb_limits()
    return int(budget) <= max_n and int(dimension) <= max_k


class BatchedPCABO:
    """`problems[b]`: ioh-like objects (`.bounds.lb/.ub`, `.meta_data.n_variables`, callable) of ONE dimension;
    `seeds[b]`: the run's seed (ExperimentRunner.py:146).  After `run()`: `x_evals[b]`, `f_evals[b]`, `current_best[b]`,
    `current_best_index[b]`, `timing` (seconds per phase, summed over iterations)."""

    def __init__(self, problems: Sequence, seeds: Sequence[int], budget: int, n_DoE: int, n_components: int = 0,
                 var_threshold: float = 0.95, acquisition_function: str = "expected_improvement",
                 maximization: bool = False, device: int = 0, num_restarts: int = 10, raw_samples: int = 512,
                 record_trace: bool = False, host_threads: int = 0, device_objective: bool = False, workers: int = 0,
                 trace_filter=None, acq_kernel: str = "group", lbfgsb_cus: int = 0, torch_threads: Optional[int] = 4,
                 gc_freeze: bool = True):
        self.problems, self.seeds = list(problems), [int(s) for s in seeds]
        self.B = len(self.problems)
        assert self.B == len(self.seeds) and self.B >= 1
        self.dimension = int(self.problems[0].meta_data.n_variables)
        assert all(int(p.meta_data.n_variables) == self.dimension for p in self.problems), "one dimension per batch"
        self.budget, self.n_DoE = int(budget), int(n_DoE) if n_DoE else self.dimension
        self.n_components, self.var_threshold, self.maximization = int(n_components), float(var_threshold), bool(maximization)
        name = {"EI": "expected_improvement", "PI": "probability_of_improvement"}.get(acquisition_function, acquisition_function)
        if name not in ("expected_improvement", "probability_of_improvement"):
            raise ValueError("Oddly defined name")
        self.acq_code = _native.ACQ_LOG_EI if name == "expected_improvement" else _native.ACQ_PI
        self.num_restarts, self.raw_samples, self.device = int(num_restarts), int(raw_samples), int(device)
        self.bounds = [np.column_stack([np.asarray(p.bounds.lb, dtype=float), np.asarray(p.bounds.ub, dtype=float)])
                       for p in self.problems]
        self.x_evals: List[List[np.ndarray]] = [[] for _ in range(self.B)]
        self.f_evals: List[List[float]] = [[] for _ in range(self.B)]
        self.current_best = [None] * self.B
        self.current_best_index = [0] * self.B
        self.k_prev = [0] * self.B
        self.lbfgsb_info = []
        self.k_hist = []                   # reduced dimension of every run, per iteration (BatchedVanillaBO: always d)
        self.retries = 0
        # failed[b]: None, or (n at failure, message).  The reference's run dies with an exception when botorch meets a NaN
        # acquisition gradient (reached by the reference's own dynamics: candidates outside the box are penalised but kept,
        # the PCA search box grows by half each time, coordinates reach 1e30 and beyond).  A single PCA_BO raises there too;
        # in a batch the run is PARKED instead: its lists stop growing at the failure (x_evals / f_evals hold what it had
        # reached), the other runs go on, and `run()` reports the failures at the end.
        self.failed = [None] * self.B
        self._frozen = [None] * self.B
        self.timing = {"pca": 0.0, "wait_score": 0.0, "init_pick": 0.0, "lbfgsb": 0.0, "tail": 0.0, "host_prep": 0.0}
        # record_trace: one entry per (run, iteration) that `trace_filter(b, n)` admits (None: all) with what the oracle
        # needs to replay that iteration from the same state - the run's numpy / torch generator states in front of the
        # iteration (in the form np.random.set_state / torch.set_rng_state take), best_f, and what the device produced
        self.record_trace, self.trace, self._trace_filter = bool(record_trace), [], trace_filter
        self._batch: Optional[_native.Batch] = None
        self._rs = self._tg = None
        self._X = None
        self._F = None                     # B x budget objective values as the device sees them (grows with the runs)
        # the runs' noise draws (numpy releases the interpreter lock while it generates) are spread over a few threads;
        # every run has its own generator, so the order in which the threads get to the runs does not matter
        # device_objective: the B candidates of an iteration are evaluated in one launch (pcabo.bbob_device; in-repo BBOB
        # problems only).  Off by default: the host evaluation keeps a run bit-identical to the same run alone.
        self._device_objective, self._dev_obj = bool(device_objective), None
        self._pool = None
        self._host_threads = int(host_threads) if host_threads else min(8, self.B)
        self._workers = int(workers)                 # 0: the library's default (pcabo_batch_set_workers)
        # the NEXT iteration's noise blocks (numpy's generator releases the interpreter lock while it fills them) are drawn on
        # the host threads while the L-BFGS-B rounds of this iteration run inside the library: same numbers from the same
        # per-run streams - nothing else draws from them after the DoE - 1.5 ms less in front of every lock-step iteration.
        # Off while per-iteration generator states are recorded (they must be the states BEFORE the draw).
        self._torch_threads, self._saved_torch_threads = torch_threads, None
        self._gc_freeze, self._gc_entered = bool(gc_freeze), False
        self._noise_ahead = {}
        self._noise_next = None            # (B, n + 1, d) block the pool threads fill for the next iteration
        # likewise the scrambled Sobol engines of the next iteration (torch's two randint draws per run, 0.06 ms each and
        # serial under the interpreter lock): built by one pool thread with THIS iteration's k as the guess while the main
        # thread sits in pcabo_batch_optimize_acqf (which releases the lock).  The draws leave each run's torch generator
        # exactly where the next iteration would take them; a run that needs botorch's retry after the call (fresh initial
        # conditions from the same generator) gets its generator put back first, a wrong guess of k likewise.
        self._engines_ahead = {}           # run -> (generator state before the draw, engine), filled during the rounds
        # acq_kernel: "group" (default: the throughput kernel k_acq_group for the L-BFGS-B rounds; a run is bit-identical to
        # PCA_BO(acq_kernel="group")) or "latency" (the per-query kernels: bit-identical to PCA_BO's default)
        # or "device" (every restart group's whole L-BFGS-B inside one kernel launch, csrc/kernels_lbfgsb.hip: no host round
        # trips, no worker threads; an evaluation order of its own) / "device-twin" (the same evaluation kernel stepped by the
        # host's L-BFGS-B, launch by launch: what "device" is compared with bit for bit)
        if acq_kernel not in ("group", "latency", "device", "device-twin"):
            raise ValueError("acq_kernel must be 'group', 'latency', 'device' or 'device-twin'")
        if acq_kernel in ("device", "device-twin") and not device_mode_covers(self.dimension, self.budget):
            # the library would serve such a run's calls host-paced - another summation order - once it leaves the device
            # optimiser's limits: one arithmetic mode per run, decided before it starts
            raise ValueError("acq_kernel=%r: a run of dimension %d with budget %d can leave the device optimiser's limits "
                             "(n <= %d, k <= %d); use 'group'" % ((acq_kernel, self.dimension, self.budget) + _native.device_lbfgsb_limits()[:2]))
        self._group_acq = acq_kernel != "latency"
        self._device_lbfgsb = {"device": 1, "device-twin": 2}.get(acq_kernel, 0)
        self._lbfgsb_cus = int(lbfgsb_cus)         # "device": the optimiser's launches confined to that many CUs (0: the whole chip)

    # ---- seeding + DoE (AbstractAlgorithm.py:310-328, AbstractBayesianOptimizer.py:142-176) --------------------------
    def start(self) -> None:
        B, d = self.B, self.dimension
        # torch's intra-op pool at the machine's core count turns the loop's small tensor operations (std, exp, multinomial on 512
        # values) into a fight between spinning OpenMP workers and the batches' own threads: main.py's default experiment ran at
        # 704 it/s instead of ~3 000 until this was capped (Algorithms.PCA_BO does the same for a single run)
        if self._torch_threads is not None and torch.get_num_threads() > int(self._torch_threads):
            self._saved_torch_threads = torch.get_num_threads()
            torch.set_num_threads(int(self._torch_threads))
        if self._gc_freeze and not self._gc_entered:       # pcabo/gcguard.py: the collector's full passes stay out of the loop
            from . import gcguard
            gcguard.enter()
            self._gc_entered = True
        self._rs = [np.random.RandomState(s) for s in self.seeds]
        # a run's torch CPU generator as the state blob torch exports, advanced by libpcabo's host helpers (pcabo/hostrng.py: same
        # numbers and consumption as torch's randint / multinomial, checked at import; a real torch.Generator otherwise)
        self._tg = [HostMT(s) for s in self.seeds]
        self._X = np.empty((B, self.budget, d))
        self._F = np.empty((B, self.budget))
        for b in range(B):
            unit = lhs_center(d, self.n_DoE, self._rs[b])
            span = self.bounds[b][:, 1] - self.bounds[b][:, 0]
            for point in span * unit + self.bounds[b][:, 0]:
                self.x_evals[b].append(point)
                self.f_evals[b].append(self.problems[b](point))
            self._assign_new_best(b)
            self._X[b, : self.n_DoE] = np.vstack(self.x_evals[b])
            self._F[b, : self.n_DoE] = self.f_evals[b]
        self._batch = _native.Batch(B, max_n=self.budget, max_d=d, max_q=max(self.raw_samples, 16), device=self.device,
                                    workers=self._workers, group_acq=self._group_acq, device_lbfgsb=self._device_lbfgsb,
                                    lbfgsb_cus=self._lbfgsb_cus)
        if self._device_objective:
            from .bbob_device import DeviceObjectives
            self._dev_obj = DeviceObjectives(self.problems, device=self.device, penalty=OOB_PENALTY)
        # (a pool of ONE thread still pays: next iteration's noise blocks and Sobol engines are made beside the L-BFGS-B phase)
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=max(1, self._host_threads))

    def _each(self, fn):
        """fn(b) for every run, on the host threads."""
        if self._pool is None or self._host_threads <= 1:
            return [fn(b) for b in range(self.B)]
        return list(self._pool.map(fn, range(self.B)))

    def _assign_new_best(self, b: int, appended: bool = False) -> None:
        """AbstractBayesianOptimizer.assign_new_best (:196-208): best = min / max of f_evals, its index searched from the previous
        best index.  `appended`: only the last value is new - the same result without walking the list (a strictly better value
        is the new best at the last index; anything else leaves both as they are)."""
        f = self.f_evals[b]
        if appended and len(f) > 1:
            new, cur = f[-1], self.current_best[b]
            if (new > cur) if self.maximization else (new < cur):
                self.current_best[b] = new
                self.current_best_index[b] = len(f) - 1
            return
        self.current_best[b] = max(f) if self.maximization else min(f)
        self.current_best_index[b] = f.index(self.current_best[b], self.current_best_index[b])

    @property
    def n(self) -> int:
        """Evaluated points of the runs that are still advancing (parked runs stopped earlier)."""
        live = [len(self.f_evals[b]) for b in range(self.B) if self.failed[b] is None]
        return max(live) if live else self.budget

    def _park(self, b: int, n: int, message: str) -> None:
        self.failed[b] = (n, message)
        reps = -(-self.budget // self.n_DoE)
        Xd = np.vstack(self.x_evals[b][: self.n_DoE])
        fd = np.array(self.f_evals[b][: self.n_DoE], dtype=float)
        self._frozen[b] = (np.tile(Xd, (reps, 1))[: self.budget], np.tile(fd, reps)[: self.budget])
        self._X[b] = self._frozen[b][0]
        self._F[b] = self._frozen[b][1]
        self.current_best[b] = float(np.min(fd) if not self.maximization else np.max(fd))
        self._batch.set_active([self.failed[i] is None for i in range(self.B)])
        warnings.warn(f"run {b} (seed {self.seeds[b]}) stopped at n = {n}: {message}", RuntimeWarning)

    # ---- one lock-step BO iteration (PCA_BO.py:178-298 for every run) ------------------------------------------------
    def iteration(self) -> None:
        for _ in self._iteration_steps():
            pass                                  # (every wait then happens inside the library call that follows the yield)

    def _iteration_steps(self):
        """The iteration as a generator: it yields wherever the next library call would wait for the device, so that ONE host
        thread can advance several batches (run_interleaved resumes a batch once its stream has drained).  Driven straight
        through (iteration()) it is the blocking iteration: same calls, same order, same results."""
        B, d, n, bt = self.B, self.dimension, self.n, self._batch
        t0 = perf_counter()
        pre = {}
        if self.record_trace:
            for b in range(B):
                if self.failed[b] is None and (self._trace_filter is None or self._trace_filter(b, n)):
                    pre[b] = {"numpy_state": self._rs[b].get_state(), "torch_state": self._tg[b].get_state().clone(),
                              "best_f": self.current_best[b]}
        self._pre_states = pre            # (kept on the object: still there when a run stops in this iteration)
        F = np.ascontiguousarray(self._F[:, :n])      # B x n (a parked run: its finite stand-in, see _park)
        # ranks as the reference forms them per run (PCA_BO.py:330-333; the penalty value repeats, so how numpy's unstable sort
        # orders ties matters): argsort along the rows of the B x n array sorts every row with the routine a 1-D array gets - the
        # same permutation, ties included (checked on 6 000 rows with repeated values; the bit-for-bit tests compare whole runs)
        ranks = np.argsort(np.argsort(-F if self.maximization else F, axis=1), axis=1).astype(np.int64) + 1
        ahead, self._noise_ahead = self._noise_ahead, {}
        # (all blocks drawn ahead: the pool threads have written them into _noise_next themselves - no copy on this thread)
        noise = self._noise_next if (len(ahead) == B and self._noise_next is not None and self._noise_next.shape == (B, n, d)) \
            else np.empty((B, n, d))
        self._noise_next = None

        def prep(b):
            fut = ahead.get(b)
            if fut is not None:
                nz = fut.result()
                if nz.shape != (n, d):
                    raise RuntimeError("noise drawn ahead is out of step with the run")
                if nz.base is not noise:
                    noise[b] = nz
            else:
                noise[b] = self._rs[b].normal(0, 1e-8, size=(n, d))                   # PCA_BO.py:376, the run's own stream
        if len(ahead) == B:                        # all drawn ahead: nothing left that a pool thread would do faster
            for b in range(B):
                prep(b)
        else:
            self._each(prep)
        t1 = perf_counter()
        bt.wpca_gp_condition_begin(self._X[:, :n], ranks, noise, self._F[:, :n], maximize=self.maximization,
                                   var_threshold=self.var_threshold, n_components=self.n_components,
                                   lengthscale=LENGTHSCALE, gp_noise=NOISE)
        # while the device runs the eigen-decompositions: the scrambled Sobol engines, with last iteration's k
        engines, saved = [None] * B, [None] * B

        built, self._engines_ahead = self._engines_ahead, {}

        def guess(b):
            if b in built:
                saved[b], engines[b] = built[b]
            elif self.k_prev[b]:
                saved[b] = self._tg[b].get_state()
                engines[b] = _init.scrambled_sobol_engine(self.k_prev[b], self._tg[b])
        for b in range(B):                # (torch's small ops do not gain from the host threads: measured slower)
            guess(b)
        yield "conditioning"
        res = bt.wpca_results()
        self.k_hist.append(np.array([r["k"] for r in res], dtype=np.int32))
        t2 = perf_counter()
        bounds = bt.acq_bounds()
        raw = [None] * B
        rawbuf = bt.raw_row_buffer(self.raw_samples)      # the runs' raw samples are drawn straight into the rows the scoring packs

        def draw(b):
            if engines[b] is not None and res[b]["k"] != self.k_prev[b]:
                self._tg[b].set_state(saved[b])                                        # wrong guess: as if never drawn
                engines[b] = None
            if engines[b] is None:
                engines[b] = _init.scrambled_sobol_engine(res[b]["k"], self._tg[b])
            self.k_prev[b] = res[b]["k"]
        for b in range(B):
            draw(b)
        raw = _native.sobol_draw_rows(engines, self.raw_samples, bt.acq_bounds_packed, rawbuf)       # all runs' points, one call
        best_f = [self.current_best[b] for b in range(B)]
        for b in range(B):
            bt.ctx[b].match_best_f_dtype(best_f[b])
        t3 = perf_counter()
        token = bt.gp_eval_begin(raw, best_f, self.maximization, self.acq_code)
        yield "scoring"
        vals, status = bt.gp_eval_end(token)
        for b in range(B):
            if self.failed[b] is None and (status[b] != 0 or not np.isfinite(vals[b]).all()):
                self._park(b, n, "GP conditioning failed (K not positive definite)" if status[b] != 0
                           else "non-finite acquisition values on the raw samples")
            if self.failed[b] is not None:
                vals[b] = np.linspace(0.0, 1.0, vals.shape[1])          # anything finite: the pick below is discarded
        t4 = perf_counter()
        pick = _init.initialize_q_batch if self.acq_code == _native.ACQ_LOG_EI else _init.initialize_q_batch_nonneg
        if self.acq_code == _native.ACQ_LOG_EI:   # all runs' Boltzmann weights at once, a run's own generator for its draw
            idx = _init.initialize_q_batch_rows(vals, self.num_restarts, self._tg,
                                                skip=[b for b in range(B) if self.failed[b] is not None])
        else:
            idx = [pick(vals[b], self.num_restarts, generator=self._tg[b]) if self.failed[b] is None
                   else np.arange(self.num_restarts) for b in range(B)]
        ics = [raw[b][idx[b]] for b in range(B)]
        t5 = perf_counter()
        if self._pool is not None and not self.record_trace and n + 1 < self.budget:
            # one task per pool thread, each drawing the blocks of its share of the runs (a submit per run cost the host thread
            # 0.7 ms per iteration at 30 runs); every run's block comes from its own stream, so the grouping changes nothing
            alive = [b for b in range(B) if self.failed[b] is None]
            nxt = self._noise_next = np.empty((B, n + 1, d)) if len(alive) == B else None
            if nxt is None:
                nxt = np.empty((B, n + 1, d))
            T = max(1, min(self._host_threads, len(alive)))
            for t in range(T):
                mine = alive[t::T]

                def draw_many(mine=mine, shape=(n + 1, d), dest=nxt):
                    out = {}
                    for b in mine:
                        dest[b] = self._rs[b].normal(0, 1e-8, shape)
                        out[b] = dest[b]
                    return out
                job = self._pool.submit(draw_many)
                for b in mine:
                    self._noise_ahead[b] = _Share(job, b)
            live = [b for b in range(B) if self.failed[b] is None and self.k_prev[b]]

            def build_engines():
                out = {}
                for b in live:
                    st = self._tg[b].get_state()
                    out[b] = (st, _init.scrambled_sobol_engine(self.k_prev[b], self._tg[b]))
                return out
            engines_job = self._pool.submit(build_engines)
        else:
            engines_job = None
        token = bt.optimize_begin(ics, bounds, best_f, self.maximization, self.acq_code, batch_limit=5, maxiter=200) \
            if self._device_lbfgsb == 1 else None
        if token is not None:             # the device-resident optimiser: one launch, collected when it has drained
            yield "optimize"
            outs, status = bt.optimize_end(token)
        else:
            outs, status = bt.optimize_acqf(ics, bounds, best_f, self.maximization, self.acq_code, batch_limit=5, maxiter=200)
        if engines_job is not None:
            # botorch's retry below draws from a run's generator: a run that needs it takes its generator back first
            built = engines_job.result()
            for b in list(built):
                if status[b] != 0 or outs[b][3]:
                    self._tg[b].set_state(built[b][0])
                    del built[b]
            self._engines_ahead = built
        for b in range(B):
            if self.failed[b] is None and status[b] != 0:
                self._park(b, n, "NaN in the acquisition gradient (botorch raises here)" if status[b] == -4
                           else f"acquisition optimisation failed (status {int(status[b])})")
        t6 = perf_counter()
        z_new, infos = [], []
        for b in range(B):
            cand, v, info, failed = outs[b]
            if self.failed[b] is not None:
                z_new.append(np.zeros(int(bt.k[b])))
                infos.append(info)
                continue
            retried = False
            if failed:       # botorch: OptimizationWarning -> one retry with freshly drawn initial conditions (this run alone)
                warnings.warn("Optimization failed in `gen_candidates_scipy`; trying again with a new set of "
                              "initial conditions.", RuntimeWarning)
                self.retries += 1
                retried = True
                c = bt.ctx[b]
                raw_b = _init.draw_sobol(bounds[b], self.raw_samples, _init.scrambled_sobol_engine(int(bt.k[b]), self._tg[b]))
                vals_b = c.acq_eval(raw_b, best_f[b], self.maximization, self.acq_code, grad=False)
                ics_b = raw_b[pick(vals_b, self.num_restarts, generator=self._tg[b])]
                cand, v, info, failed = c.optimize_acqf(ics_b, bounds[b], best_f[b], self.maximization, self.acq_code,
                                                        batch_limit=5, maxiter=200)
                ics[b] = ics_b
            best = int(np.argmax(v))
            z_new.append(cand[best])
            infos.append(info)
            if b in pre:
                self.trace.append({"b": b, "n": n, "k": int(bt.k[b]), "ic_idx": np.asarray(idx[b]).copy(), "ics": ics[b].copy(),
                                   "cands": cand.copy(), "vals": v.copy(), "info": info.copy(), "chosen": best,
                                   "retried": retried, **pre[b]})
        self.lbfgsb_info.append(infos)
        bt.inverse_map_begin(z_new)
        yield "inverse map"
        X_new = bt.inverse_map_end()
        f_dev = None
        if self._dev_obj is not None and not self.maximization:
            f_dev, raw_dev, oob_dev = self._dev_obj.evaluate(X_new)
        for b in range(B):
            if self.failed[b] is not None:
                continue
            new_x = X_new[b].copy()
            if f_dev is not None:
                new_f = float(f_dev[b])
                if not oob_dev[b]:               # keep the problem's own record (what the Analyzer rows are written from)
                    p = self.problems[b]
                    p.evaluations += 1
                    p.best_raw = min(p.best_raw, float(raw_dev[b]))
                    p.log.append((float(raw_dev[b]), new_x.copy()))
                self.x_evals[b].append(new_x)
                self.f_evals[b].append(new_f)
                self._X[b, n] = new_x
                self._F[b, n] = new_f
                self._assign_new_best(b, appended=True)
                continue
            outside = not np.all(new_x >= self.bounds[b][:, 0]) or not np.all(new_x <= self.bounds[b][:, 1])
            # out-of-box candidates are not evaluated; they cost budget and a fixed penalty (PCA_BO.py:260-263)
            new_f = (-OOB_PENALTY if self.maximization else OOB_PENALTY) if outside else self.problems[b](new_x)
            self.x_evals[b].append(new_x)
            self.f_evals[b].append(new_f)
            self._X[b, n] = new_x
            self._F[b, n] = new_f
            self._assign_new_best(b, appended=True)
        t7 = perf_counter()
        tm = self.timing
        tm["host_prep"] += t1 - t0
        tm["pca"] += t2 - t1
        tm["wait_score"] += t4 - t2
        tm["init_pick"] += t5 - t4
        tm["lbfgsb"] += t6 - t5
        tm["tail"] += t7 - t6

    def finish(self) -> None:
        if getattr(self, "_saved_torch_threads", None) is not None:
            torch.set_num_threads(self._saved_torch_threads)
            self._saved_torch_threads = None
        if getattr(self, "_gc_entered", False):
            from . import gcguard
            gcguard.leave()
            self._gc_entered = False
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None
        if self._dev_obj is not None:
            self._dev_obj.close()
            self._dev_obj = None
        if self._batch is not None:
            self._batch.close()
            self._batch = None

    def run(self) -> None:
        """All iterations of all runs.  Runs that failed on the way are listed in `failed` (the reference's run would
        have ended with an exception at that point); everything else is complete."""
        try:
            self.start()
            while self.n < self.budget:
                self.iteration()
        finally:
            self.finish()


class BatchedVanillaBO(BatchedPCABO):
    """B runs of the reference's Vanilla_BO (Vanilla_BO.py:39-301) in lock-step: the PCA_BO loop without the PCA - the exact GP
    on the raw d-dimensional points (Normalize switched off: identity bounds, Vanilla_BO.py:188-194), the acquisition optimised
    inside the problem's box (:206-213), every candidate evaluated (no out-of-box rule).  Same device calls as BatchedPCABO minus
    rows A-C (pcabo_batch_gp_condition_begin); every run has its own torch.Generator seeded like the reference's global one and
    takes, bit for bit, the path `Algorithms.Vanilla_BO` takes alone with the same evaluation kernels."""

    def start(self) -> None:
        super().start()
        d = self.dimension
        self._identity = np.vstack([np.zeros(d), np.ones(d)])
        self._boxes = [np.ascontiguousarray(self.bounds[b].T, dtype=np.float64) for b in range(self.B)]     # 2 x d each
        self._boxes_packed = np.ascontiguousarray(np.stack([bx.ravel() for bx in self._boxes]))             # B x [lo(d), hi(d)]
        self.k_prev = [d] * self.B

    def _iteration_steps(self):
        B, d, n, bt = self.B, self.dimension, self.n, self._batch
        t0 = perf_counter()
        pre = {}
        if self.record_trace:
            for b in range(B):
                if self.failed[b] is None and (self._trace_filter is None or self._trace_filter(b, n)):
                    pre[b] = {"numpy_state": self._rs[b].get_state(), "torch_state": self._tg[b].get_state().clone(),
                              "best_f": self.current_best[b]}
        self._pre_states = pre
        t1 = perf_counter()
        bt.gp_condition_begin(self._X[:, :n], self._F[:, :n], norm_bounds=self._identity, lengthscale=LENGTHSCALE, gp_noise=NOISE)
        # the scrambled Sobol engines (dimension d, always): built during the last iteration's optimiser phase, or now - while the
        # device conditions the GPs
        built, self._engines_ahead = self._engines_ahead, {}
        bounds = self._boxes
        rawbuf = bt.raw_row_buffer(self.raw_samples)
        engines = [built[b][1] if b in built else _init.scrambled_sobol_engine(d, self._tg[b]) for b in range(B)]
        raw = _native.sobol_draw_rows(engines, self.raw_samples, self._boxes_packed, rawbuf)
        best_f = [self.current_best[b] for b in range(B)]
        for b in range(B):
            bt.ctx[b].match_best_f_dtype(best_f[b])
        t3 = t2 = perf_counter()
        token = bt.gp_eval_begin(raw, best_f, self.maximization, self.acq_code)
        yield "scoring"
        vals, status = bt.gp_eval_end(token)
        for b in range(B):
            if self.failed[b] is None and (status[b] != 0 or not np.isfinite(vals[b]).all()):
                self._park(b, n, "GP conditioning failed (K not positive definite)" if status[b] != 0
                           else "non-finite acquisition values on the raw samples")
            if self.failed[b] is not None:
                vals[b] = np.linspace(0.0, 1.0, vals.shape[1])
        t4 = perf_counter()
        pick = _init.initialize_q_batch if self.acq_code == _native.ACQ_LOG_EI else _init.initialize_q_batch_nonneg
        if self.acq_code == _native.ACQ_LOG_EI:
            idx = _init.initialize_q_batch_rows(vals, self.num_restarts, self._tg,
                                                skip=[b for b in range(B) if self.failed[b] is not None])
        else:
            idx = [pick(vals[b], self.num_restarts, generator=self._tg[b]) if self.failed[b] is None
                   else np.arange(self.num_restarts) for b in range(B)]
        ics = [raw[b][idx[b]] for b in range(B)]
        t5 = perf_counter()
        engines_job = None
        if self._pool is not None and not self.record_trace and n + 1 < self.budget:
            live = [b for b in range(B) if self.failed[b] is None]

            def build_engines():
                out = {}
                for b in live:
                    st = self._tg[b].get_state()
                    out[b] = (st, _init.scrambled_sobol_engine(d, self._tg[b]))
                return out
            engines_job = self._pool.submit(build_engines)
        token = bt.optimize_begin(ics, bounds, best_f, self.maximization, self.acq_code, batch_limit=5, maxiter=200) \
            if self._device_lbfgsb == 1 else None
        if token is not None:
            yield "optimize"
            outs, status = bt.optimize_end(token)
        else:
            outs, status = bt.optimize_acqf(ics, bounds, best_f, self.maximization, self.acq_code, batch_limit=5, maxiter=200)
        if engines_job is not None:
            built = engines_job.result()
            for b in list(built):
                if status[b] != 0 or outs[b][3]:          # botorch's retry draws from the run's generator first
                    self._tg[b].set_state(built[b][0])
                    del built[b]
            self._engines_ahead = built
        for b in range(B):
            if self.failed[b] is None and status[b] != 0:
                self._park(b, n, "NaN in the acquisition gradient (botorch raises here)" if status[b] == -4
                           else f"acquisition optimisation failed (status {int(status[b])})")
        t6 = perf_counter()
        infos = []
        for b in range(B):
            cand, v, info, failed = outs[b]
            infos.append(info)
            if self.failed[b] is not None:
                continue
            retried = False
            if failed:
                warnings.warn("Optimization failed in `gen_candidates_scipy`; trying again with a new set of "
                              "initial conditions.", RuntimeWarning)
                self.retries += 1
                retried = True
                c = bt.ctx[b]
                raw_b = _init.draw_sobol(bounds[b], self.raw_samples, _init.scrambled_sobol_engine(d, self._tg[b]))
                vals_b = c.acq_eval(raw_b, best_f[b], self.maximization, self.acq_code, grad=False)
                ics_b = raw_b[pick(vals_b, self.num_restarts, generator=self._tg[b])]
                cand, v, info, failed = c.optimize_acqf(ics_b, bounds[b], best_f[b], self.maximization, self.acq_code,
                                                        batch_limit=5, maxiter=200)
                ics[b] = ics_b
                infos[-1] = info
            best = int(np.argmax(v))
            if b in pre:
                self.trace.append({"b": b, "n": n, "k": d, "ic_idx": np.asarray(idx[b]).copy(), "ics": ics[b].copy(),
                                   "cands": cand.copy(), "vals": v.copy(), "info": info.copy(), "chosen": best,
                                   "retried": retried, **pre[b]})
            new_x = np.asarray(cand[best], dtype=np.float64).ravel().copy()
            new_f = self.problems[b](new_x)                # (Vanilla_BO.py:221-232: the candidate lies in the box and is evaluated)
            self.x_evals[b].append(new_x)
            self.f_evals[b].append(new_f)
            self._X[b, n] = new_x
            self._F[b, n] = new_f
            self._assign_new_best(b, appended=True)
        self.lbfgsb_info.append(infos)
        t7 = perf_counter()
        tm = self.timing
        tm["host_prep"] += t1 - t0
        tm["pca"] += t2 - t1
        tm["wait_score"] += t4 - t3
        tm["init_pick"] += t5 - t4
        tm["lbfgsb"] += t6 - t5
        tm["tail"] += t7 - t6


def workers_for(side_by_side: int) -> int:
    """Gang threads per batch when `side_by_side` batches of this process advance at once: the workers spin, and a GPU of
    a shared node comes with ~16 cores - 8 for one batch, 4 each for two, never fewer than 2."""
    return max(2, 8 // max(1, int(side_by_side)))


def run_side_by_side(runners: Sequence["BatchedPCABO"], started: bool = False) -> None:
    """Advance several batches at once, one host thread each (the library calls release the interpreter lock): while one
    batch is in its L-BFGS-B rounds - host-paced, the GPU lightly used - another one has the device for its wPCA /
    conditioning / scoring launches and the interpreter for its bookkeeping.  The batches are independent (their own
    contexts, streams and generators), so every run is bit-identical to the same run in any other grouping.  An
    exception of one batch is re-raised after all of them have ended."""
    import threading
    errors: List[BaseException] = []

    def drive(r: "BatchedPCABO") -> None:
        try:
            if not started:
                r.start()
            while r.n < r.budget:
                r.iteration()
        except BaseException as e:      # noqa: BLE001 - handed to the caller below
            errors.append(e)
        finally:
            if not started:
                r.finish()

    if len(runners) == 1:
        drive(runners[0])
    else:
        # (the interpreter's switch interval makes no difference here - 5 ms, 0.2 ms, 0.02 ms measured on 4 x 30 runs: what the
        # threads wait for is each other's bookkeeping, see run_interleaved)
        threads = [threading.Thread(target=drive, args=(r,), name=f"pcabo-batch-{i}") for i, r in enumerate(runners)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if errors:
        raise errors[0]


LAST_INTERLEAVE_STATS: dict = {}      # of the last run_interleaved call: how long the one host thread was busy, per segment


def run_interleaved(runners: Sequence["BatchedPCABO"], started: bool = False) -> None:
    """Advance several batches from ONE host thread: every batch is run as far as its next wait for the device
    (_iteration_steps), then the next batch gets the interpreter; a batch is resumed once its stream has drained
    (pcabo_batch_busy).  Made for acq_kernel = "device", where the long wait of an iteration - the L-BFGS-B phase - is a single
    launch: the batches' host work (ranks, noise, Sobol engines, picks, bookkeeping) runs back to back without the threads'
    fight for the interpreter lock, while up to len(runners) optimisation launches share the GPU.  Batches are independent, so
    every run is bit-identical to the same run in any other grouping or schedule."""
    if not started:
        for r in runners:
            r.start()
    steps = {id(r): None for r in runners}
    live = list(runners)
    stats = LAST_INTERLEAVE_STATS
    stats.clear()
    stats.update({"host_busy_seconds": 0.0, "resumptions": 0, "segments": {}})
    t_start = perf_counter()
    try:
        while live:
            for r in list(live):
                g = steps[id(r)]
                if g is None:
                    if r.n >= r.budget:
                        live.remove(r)
                        continue
                    g = steps[id(r)] = r._iteration_steps()
                elif r._batch.busy():
                    continue
                t0 = perf_counter()
                try:
                    where = next(g)
                except StopIteration:
                    steps[id(r)] = None
                    where = "end"
                dt = perf_counter() - t0
                stats["host_busy_seconds"] += dt
                stats["resumptions"] += 1
                seg = stats["segments"].setdefault(where, [0.0, 0])
                seg[0] += dt
                seg[1] += 1
        stats["wall_seconds"] = perf_counter() - t_start
    finally:
        for g in steps.values():
            if g is not None:
                g.close()
        if not started:
            for r in runners:
                r.finish()


def bench_block(device: int, B: int, fid: int, dim: int, budget_factor: int = 10, doe_factor: float = 3.0,
                sub_batches: int = 1, workers: int = 0, acq_kernel: str = "group", schedule: str = "threads",
                lbfgsb_cus: int = 0, device_objective: bool = False, algorithm: str = "pca") -> dict:
    """Aggregate BO iterations / second of B runs (instances 0..B-1 of one BBOB function and dimension, seeds per
    ExperimentRunner.py:146) advancing together on one GPU - as one lock-step batch, or as `sub_batches` lock-step batches
    side by side (run_side_by_side); DoE and set-up untimed."""
    from .bbob import BBOBProblem
    budget, n_doe = budget_factor * dim + 50, int(doe_factor * dim)
    S = max(1, min(int(sub_batches), B))
    subs = []
    for t in range(S):
        inst = list(range(t, B, S))
        subs.append((BatchedVanillaBO if algorithm == "vanilla" else BatchedPCABO)(
            [BBOBProblem(fid, i, dim) for i in inst], [1000 * fid + 10 * dim + i for i in inst], budget, n_doe,
                                 device=device, workers=workers or (workers_for(S) if S > 1 else 0), host_threads=max(1, 8 // S),
                                 acq_kernel=acq_kernel, lbfgsb_cus=lbfgsb_cus, device_objective=device_objective))
    for r in subs:
        r.start()
    torch.cuda.synchronize()
    t0 = perf_counter()
    try:
        if schedule.startswith("interleaved") and len(schedule) > len("interleaved"):
            # "interleaved2", "interleaved3": that many host threads, each interleaving its share of the batches
            import threading
            T = max(1, min(int(schedule[len("interleaved"):]), S))
            errs: List[BaseException] = []

            def drive(part):
                try:
                    run_interleaved(part, started=True)
                except BaseException as e:      # noqa: BLE001
                    errs.append(e)
            th = [threading.Thread(target=drive, args=(subs[t::T],)) for t in range(T)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if errs:
                raise errs[0]
        else:
            (run_interleaved if schedule == "interleaved" else run_side_by_side)(subs, started=True)
        torch.cuda.synchronize()
        dt = perf_counter() - t0
    finally:
        for r in subs:
            r.finish()
    iters = sum(len(f) - n_doe for r in subs for f in r.f_evals)           # (parked runs count what they completed)
    phases = {k: sum(r.timing[k] for r in subs) / S for k in subs[0].timing}
    best = [None] * B
    for t, r in enumerate(subs):
        for j, i in enumerate(range(t, B, S)):
            best[i] = float(r.current_best[j])
    extra = {"interleave": {k: (v if not isinstance(v, dict) else {a: [round(b[0], 3), b[1]] for a, b in v.items()})
                            for k, v in LAST_INTERLEAVE_STATS.items()}} if schedule == "interleaved" else {}
    extra["schedule"] = schedule
    extra["device_objective"] = bool(device_objective)
    extra["algorithm"] = algorithm
    # algorithmic bytes of the L-BFGS-B evaluations (SURVEY.md 8d per-unit figures: one value+gradient evaluation of a restart
    # group reads the triangles of R and of its transpose, the normalised points twice and alpha): evaluations the optimisers
    # report x bytes at that iteration's (n, k)
    ev_bytes, evals = 0.0, 0
    for r in subs:
        for it, infos in enumerate(r.lbfgsb_info):
            n_it = n_doe + it
            for b, info in enumerate(infos):
                kb = int(r.k_hist[it][b]) if it < len(r.k_hist) else dim
                nf = int(np.asarray(info)[:, 1].sum())
                evals += nf
                ev_bytes += nf * (2 * 4.0 * n_it * (n_it + 1) + 2 * 8.0 * n_it * kb + 8.0 * n_it)
    extra["lbfgsb_group_evaluations"] = evals
    extra["lbfgsb_algorithmic_bytes"] = ev_bytes
    return {**extra, "runs": B, "sub_batches": S, "function": fid, "dimension": dim, "budget": budget, "n_DoE": n_doe,
            "aggregate_bo_iterations_per_s": iters / dt, "seconds": dt, "bo_iterations": iters,
            "ms_per_lockstep_iteration": 1e3 * dt / (budget - n_doe), "host_phase_seconds": phases,
            "retries": sum(r.retries for r in subs), "failed_runs": sum(f is not None for r in subs for f in r.failed), "best_f": best,
            "note": "B runs of configs[1]'s cell advancing in lock-step through pcabo_batch_* (one launch sequence for rows "
                    "A-H of all runs, shared acquisition launches for the L-BFGS-B rounds); one Python host thread per sub-batch"}
