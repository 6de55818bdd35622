"""Experiment runner: many independent BO runs over (algorithm, function, dimension, instance).

Same constructor and `run_experiment()` as the reference's runner
(/root/reference/Algorithms/Experiment/ExperimentRunner.py:23-200), so the reference's `main.py` works unchanged
with this package first on PYTHONPATH.  Differences, all on the far side of the hot path:

  * run-level data parallelism: under `torch.distributed.run` (RANK / WORLD_SIZE / LOCAL_RANK in the
    environment) every rank takes a balanced share of the run list (`pcabo.sharding.assign_runs`, longest first) and
    drives its own GPU (`device=LOCAL_RANK`); ranks write into `<experiment>-rank<r>` folders, no collective is
    needed.  A single process does all runs on device 0, exactly like the reference.
  * `ioh` is optional: when it imports, problems come from `ioh.iohcpp.suite.BBOB` and the real `Analyzer` logs
    them; otherwise `pcabo.bbob` (f15 / f20, pinned by the reference's data) and `pcabo.iohlog.Analyzer`
    (same on-disk layout) are used.
"""
from __future__ import annotations

import os
from time import time
from typing import List, Optional

from numpy.linalg import norm

from Algorithms import Vanilla_BO
from Algorithms import PCA_BO
from pcabo import sharding as _sharding

try:                                           # pragma: no cover - ioh is absent from the build image
    from ioh.iohcpp.suite import BBOB as _IohBBOB
    from ioh.iohcpp.logger import Analyzer as _IohAnalyzer
    from ioh.iohcpp.logger.property import RAWYBEST
    from ioh.iohcpp.logger.trigger import ALWAYS
    HAVE_IOH = True
except ImportError:
    HAVE_IOH = False
    RAWYBEST = "raw_y_best"
    ALWAYS = "always"

try:
    from tqdm.auto import tqdm
except ImportError:                            # pragma: no cover
    tqdm = None


class _NoBar:
    def __init__(self, *a, **k): pass
    def __enter__(self): return self
    def __exit__(self, *a): return False
    def update(self, n=1): pass
    def set_description(self, s): pass
    def write(self, s): print(s)
    def close(self): pass


DEVICE_BATCHES_AT_ONCE = 8       # lock-step batches of the device-resident optimiser one host thread interleaves at most


def device_batch_plan(n_runs: int):
    """Batches for the device-resident optimiser of ONE dimension: (runs per batch, batches advancing at once) - up to four batches
    of at least 30 runs at a time, none larger than 120 runs (45 -> 1 x 45; 90 -> 3 x 30; 600 -> 8 x 75, four at a time).  The
    runner lets the groups of different dimensions advance together up to DEVICE_BATCHES_AT_ONCE batches (_run_batched)."""
    nb = min(4, max(1, n_runs // 30))
    rounds = -(-n_runs // (120 * nb))
    return -(-n_runs // (nb * rounds)), nb


def merge_batch_groups(groups, at_once: int):
    """Neighbouring groups of batches joined while a joined group holds at most `at_once` batches (order kept, no batch split)."""
    merged = []
    for g in groups:
        if merged and len(merged[-1]) + len(g) <= at_once:
            merged[-1] = merged[-1] + list(g)
        else:
            merged.append(list(g))
    return merged


def resolve_arithmetic_mode(batch_acq_kernel: str, batched: int, dim: int, runs_of_dim_in_experiment: int, budget: int) -> str:
    """The arithmetic mode of every run of dimension `dim` - "latency" (per-query kernels), "group" (restart-group kernel, host-paced
    L-BFGS-B) or "device" (device-resident L-BFGS-B).  The three sum in different orders (~1e-15 relative apart), and L-BFGS-B turns
    that into another trajectory, so the mode is a property of the EXPERIMENT: it is taken from the experiment's own description -
    the request, the dimension, the runs of that dimension over ALL ranks, the budget - and never from this rank's share, the
    number of GPUs or the way the runs are grouped into batches.  One (function, instance, seed) therefore writes the same rows on
    one GPU and on eight; the mode goes into the IOHprofiler experiment attributes (`arithmetic_mode`).

      not batched            -> "latency"  (Algorithms.PCA_BO / Vanilla_BO, one run after the other: the reference's own loop)
      "group" / "latency"    -> as asked
      "device"               -> "device" where the device optimiser covers every iteration of the run (pcabo.batchrun.device_mode_covers:
                                budget <= 512, d <= 40), "group" elsewhere - per dimension, decided before the first run starts
      "auto"                 -> "device" for 20 <= d <= 40 when the experiment holds >= 30 runs of the dimension (measured, DESIGN.md
                                section 7: 30 runs 1 602 it/s against 1 345 host-paced at d = 40, 2 061 / 1 797 at d = 20, more with more
                                runs; at d = 10 the host paces the rounds faster than one wave steps them), "group" otherwise."""
    from pcabo.batchrun import device_mode_covers
    if batched <= 1:
        return "latency"
    if batch_acq_kernel in ("group", "latency"):
        return batch_acq_kernel
    if batch_acq_kernel == "device":
        return "device" if device_mode_covers(dim, budget) else "group"
    if batch_acq_kernel == "auto":
        return "device" if runs_of_dim_in_experiment >= 30 and 20 <= dim <= 40 and device_mode_covers(dim, budget) else "group"
    raise ValueError("batch_acq_kernel must be 'group', 'latency', 'device' or 'auto'")


def split_evenly(cell, batched: int, side_by_side: int):
    """Divide the runs of one dimension over lock-step batches of AT MOST `batched` runs each, as evenly as possible and in
    order; the number of batches is rounded up to a multiple of `side_by_side` so that no batch advances with nothing
    beside it (90 runs, batched=45, side_by_side=2 -> 45 + 45; 119 runs, batched=30 -> 4 x 30 (29); 30 runs, batched=30
    -> 15 + 15; never more batches than runs).  `batched` bounds what scales with the batch: device memory, the launch
    table of a worker thread and the rounds every run waits for the slowest restart group of its batch."""
    n = len(cell)
    if n == 0:
        return []
    side_by_side, batched = max(1, int(side_by_side)), max(1, int(batched))
    nb = -(-n // batched)                                   # ceil: `batched` is an upper bound
    nb = min(n, -(-nb // side_by_side) * side_by_side)
    cuts = [n * i // nb for i in range(nb + 1)]
    return [cell[cuts[i]:cuts[i + 1]] for i in range(nb)]


class ExperimentRunner:
    """Class to run and manage experiments comparing Vanilla BO and PCA-BO algorithms."""

    def __init__(self, algorithms: List[str], dimensions: List[int], problem_ids: List[int], num_runs: int = 30,
                 budget_factor: int = 10, doe_factor: float = 3.0, root_dir: str = os.getcwd(),
                 experiment_name: str = "experiment", acquisition_function: str = "expected_improvement",
                 pca_components: Optional[int] = None, var_threshold: float = 0.95, verbose: bool = False,
                 progress: bool = True, batched: int = 0, side_by_side: int = 2, batch_acq_kernel: str = "group"):
        self.algorithms = algorithms
        self.dimensions = dimensions
        self.problem_ids = problem_ids
        self.num_runs = num_runs
        self.budget_factor = budget_factor
        self.doe_factor = doe_factor
        self.root_dir = root_dir
        self.experiment_name = experiment_name
        self.acquisition_function = acquisition_function
        self.pca_components = pca_components
        self.var_threshold = var_threshold
        self.verbose = verbose
        self.progress = progress and tqdm is not None
        # batched > 1 (not in the reference): the PCA_BO runs of this rank that share a dimension advance in lock-step,
        # about `batched` at a time, through pcabo.batchrun (one launch sequence for all of them per phase).  Same runs, same
        # files: a run's rows are written once its batch has finished.  Needs the in-repo BBOB problems (no ioh logger).
        self.batched = int(batched)
        # side_by_side: that many lock-step batches advance at once, one host thread each (pcabo.batchrun.run_side_by_side):
        # one batch's host-paced L-BFGS-B rounds overlap the other's launches and bookkeeping.  Same runs, same numbers.
        self.side_by_side = max(1, int(side_by_side))
        # pcabo.batchrun.BatchedPCABO(acq_kernel=...): "group" (host-paced L-BFGS-B rounds, the default), "latency", "device"
        # (device-resident L-BFGS-B: pays from ~30 runs in flight, e.g. batched=75, side_by_side=4) or "auto" (per dimension of the
        # EXPERIMENT: resolve_arithmetic_mode above - the same answer on every rank of any world size)
        self.batch_acq_kernel = batch_acq_kernel
        self.arithmetic_modes = {dim: resolve_arithmetic_mode(batch_acq_kernel, self.batched, dim, len(problem_ids) * num_runs,
                                                              budget_factor * dim + 50) for dim in dimensions}

        self.triggers = [ALWAYS]
        self.logger_properties = [RAWYBEST]
        self.instances = range(self.num_runs)
        self.doe_params = {"criterion": "center", "iterations": 1000}

        self.rank = int(os.environ.get("RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        # one process per GPU is the normal launch; with more local ranks than GPUs the ranks share the devices
        # round-robin (independent runs from several processes overlap well on one GPU: a single run keeps it busy
        # for ~10 us out of every ~25 us round)
        from pcabo import _native
        ndev = _native.device_count()
        local = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = local % ndev if ndev > 0 else local
        self.results = []                      # one dict per finished run (this rank)
        self.failed_runs = []                  # batched mode: runs that stopped early, with the reason

    # ---- run list and its shard -----------------------------------------------------------------------------------
    def _my_runs(self):
        """(problem_id, dim, instance) triples of this rank, in suite order (ExperimentRunner.py:90,131)."""
        runs = _sharding.enumerate_runs(self.problem_ids, self.dimensions, self.num_runs)
        if self.world_size == 1:
            return runs
        mine = set(_sharding.assign_runs(runs, self.world_size, budget_factor=self.budget_factor,
                                          doe_factor=self.doe_factor)[self.rank])
        return [r for r in runs if r in mine]

    def _folder(self, algorithm: str) -> str:
        base = f"{algorithm}-{self.experiment_name}"
        return base if self.world_size == 1 else f"{base}-rank{self.rank}"

    def _make_logger(self, algorithm: str):
        kw = dict(root=self.root_dir, folder_name=self._folder(algorithm), algorithm_name=algorithm,
                  algorithm_info=f"A {algorithm}-BO Implementation.", store_positions=True)
        if HAVE_IOH:                           # pragma: no cover
            return _IohAnalyzer(triggers=self.triggers, additional_properties=self.logger_properties, **kw)
        from pcabo.iohlog import Analyzer
        return Analyzer(**kw)

    def _problems(self, logger):
        """Yield attached problems for this rank's runs."""
        mine = self._my_runs()
        if HAVE_IOH:                           # pragma: no cover
            wanted = set(mine)
            suite = _IohBBOB(problem_ids=self.problem_ids, dimensions=self.dimensions, instances=self.instances)
            suite.attach_logger(logger)
            for problem in suite:
                md = problem.meta_data
                if (md.problem_id, md.n_variables, md.instance) in wanted:
                    yield problem
            suite.detach_logger()
            return
        from pcabo.bbob import BBOBProblem
        from pcabo.iohlog import LoggedProblem
        for pid, dim, inst in mine:
            yield LoggedProblem(BBOBProblem(pid, inst, dim), logger)

    def _run_batched(self, logger, ebar, algorithm: str = "pca") -> None:
        """This rank's runs of one algorithm, `self.batched` runs of one dimension at a time in lock-step (pcabo.batchrun:
        BatchedPCABO / BatchedVanillaBO), `self.side_by_side` such batches at once."""
        from pcabo.batchrun import BatchedPCABO, BatchedVanillaBO, run_interleaved, run_side_by_side, workers_for
        Driver = BatchedVanillaBO if algorithm == "vanilla" else BatchedPCABO
        from pcabo.bbob import BBOBProblem
        from pcabo.iohlog import LoggedProblem
        mine = self._my_runs()
        groups = []                          # lists of (dim, runs of one batch, kernel): the batches of a list advance together
        device_groups = []                   # the same for the device-resident optimiser, merged across dimensions below
        for dim in sorted({r[1] for r in mine}, key=self.dimensions.index):
            cell = [r for r in mine if r[1] == dim]
            kernel = self.arithmetic_modes[dim]              # (a property of the experiment, not of this rank's share)
            if kernel == "device" and self.batch_acq_kernel == "auto":
                # the optimiser on the device: up to four batches of >= 30 runs of a dimension interleaved on one host thread (bigger
                # batches win: DESIGN.md section 7), the dimensions' groups merged below.  How this rank groups its runs changes no number: within a
                # mode a run is bit-identical in any batch (tests/test_gpu_batch.py, tests/test_gpu_device_lbfgsb.py)
                per, nb = device_batch_plan(len(cell))
                parts = [cell[i:i + per] for i in range(0, len(cell), per)]
                device_groups += [[(dim, part, kernel) for part in parts[i:i + nb]] for i in range(0, len(parts), nb)]
                continue
            # the runs of a dimension are divided EVENLY over a multiple of `side_by_side` batches of about `batched` runs
            # (a lone last batch would advance with nothing beside it; larger batches amortise the rounds of the slowest
            # restart better: 90 runs with batched=30, side_by_side=2 go as 2 x 45 rather than 30 + 30 | 30)
            parts = split_evenly(cell, self.batched, self.side_by_side)
            groups += [[(dim, part, kernel) for part in parts[i:i + self.side_by_side]]
                       for i in range(0, len(parts), self.side_by_side)]
        # device mode: the groups of different dimensions advance TOGETHER while they fit DEVICE_BATCHES_AT_ONCE batches on the one
        # host thread - with several hundred runs in flight the CUs decide (a work-group per restart group holds a CU), and more
        # queued launches fill the gaps the slowest groups of a batch leave (one MI355X, f15 d = 40: 240 runs as 4 x 60 5 400 it/s,
        # 480 runs as 8 x 60 6 830; EXPERIMENTS.md R4.4)
        groups += merge_batch_groups(device_groups, DEVICE_BATCHES_AT_ONCE)
        import torch
        saved_threads = torch.get_num_threads()      # (see BatchedPCABO.start: the loop's small tensor operations and torch's
        if saved_threads > 4:                        # intra-op pool at the machine's core count do not get along)
            torch.set_num_threads(4)
        try:
            self._run_batch_groups(groups, Driver, algorithm, logger, ebar, run_interleaved, run_side_by_side, workers_for,
                                   BBOBProblem, LoggedProblem)
        finally:
            torch.set_num_threads(saved_threads)

    def _run_batch_groups(self, groups, Driver, algorithm, logger, ebar, run_interleaved, run_side_by_side, workers_for,
                          BBOBProblem, LoggedProblem) -> None:
        for group in groups:
            jobs = []
            for dim, chunk, kernel in group:
                probs = [BBOBProblem(pid, inst, dim) for pid, _, inst in chunk]
                budget, n_doe = self.budget_factor * dim + 50, int(self.doe_factor * dim)
                seeds = [1000 * pid + 10 * dim + inst for pid, _, inst in chunk]
                runner = Driver(probs, seeds, budget, n_doe, n_components=self.pca_components or 0,
                                      var_threshold=self.var_threshold, acquisition_function=self.acquisition_function,
                                      device=self.device, workers=workers_for(len(group)) if len(group) > 1 else 0,
                                      host_threads=max(1, 8 // len(group)), acq_kernel=kernel)
                jobs.append((dim, chunk, probs, n_doe, runner))
            kernel = group[0][2]
            start_time = time()
            # "device": every batch's L-BFGS-B phase is one launch - one host thread interleaves the batches of the group
            # (pcabo.batchrun.run_interleaved); otherwise a host thread per batch
            (run_interleaved if kernel == "device" else run_side_by_side)([j[4] for j in jobs])
            elapsed = (time() - start_time) / sum(len(j[1]) for j in jobs)          # a run's share of its group of batches
            # the reference's three phase timers (PCA_BO.py:65) as FRACTIONS of the run's `time`: a batch's phase clocks are wall
            # clock differences across its waits, and while it waits the host advances the other batches of the group (interleaved)
            # or shares the cores with them (threads) - taken as seconds they would add up to several times `time`.  The
            # conditioning is enqueued together with the wPCA ("pca"); its wait falls into the optimiser's time as in the
            # reference, where gpytorch factors K lazily inside optimize_acqf
            def _fractions(t):
                total = sum(t.values()) or 1.0
                return {"pca": elapsed * (t["host_prep"] + t["pca"]) / total, "SingleTaskGP": 0.0,
                        "optimize_acqf": elapsed * (t["wait_score"] + t["init_pick"] + t["lbfgsb"]) / total}
            shares = {id(j[4]): _fractions(j[4].timing) for j in jobs}
            if algorithm == "vanilla":       # Vanilla_BO.TIME_PROFILES: the enqueue of the conditioning is its "SingleTaskGP"
                shares = {key: {"SingleTaskGP": v["pca"], "optimize_acqf": v["optimize_acqf"]} for key, v in shares.items()}
            for dim, chunk, probs, n_doe, runner in jobs:
                for b, (pid, _, inst) in enumerate(chunk):
                    replay = LoggedProblem(BBOBProblem(pid, inst, dim), logger)     # the run's rows, in its own order
                    for _, x in probs[b].log:
                        replay(x)
                    logger.set_run_attribute("time", elapsed)
                    for name, seconds in shares[id(runner)].items():
                        logger.set_run_attribute(f"{name}_time", seconds)
                    done = len(runner.f_evals[b]) - n_doe
                    self.results.append({"algorithm": algorithm, "problem_id": pid, "dim": dim, "instance": inst,
                                         "best": min(runner.f_evals[b]), "time": elapsed, "iterations": done,
                                         **shares[id(runner)]})
                    if runner.failed[b] is not None:
                        # the reference's run ends with an exception here (botorch raises on a NaN acquisition gradient) and
                        # takes the experiment with it; in a batch the other runs finish and the failure is reported
                        self.failed_runs.append({"problem_id": pid, "dim": dim, "instance": inst,
                                                 "n": runner.failed[b][0], "error": runner.failed[b][1]})
                    ebar.update(1)

    # ---- the experiment --------------------------------------------------------------------------------------------
    def run_experiment(self) -> None:
        total_runs = len(self.algorithms) * len(self.problem_ids) * len(self.dimensions) * self.num_runs
        my_total = len(self.algorithms) * len(self._my_runs())
        if self.rank == 0:
            print(f"\nRunning {total_runs} experiments ({len(self.algorithms)} algorithms × "
                  f"{len(self.dimensions)} dimensions × {len(self.problem_ids)} problems × {self.num_runs} runs)"
                  + (f" on {self.world_size} GPUs ({my_total} on rank 0)" if self.world_size > 1 else "") + "\n")
        bar = tqdm if self.progress else _NoBar

        with bar(total=my_total, position=0, desc="Total Progress") as ebar:
            for algorithm in self.algorithms:
                if algorithm not in ("vanilla", "pca"):
                    raise ValueError(f"Invalid algorithm name: '{algorithm}'")
                logger = self._make_logger(algorithm)
                # (the reference sets the first three, then REPLACES them by the PCA pair for "pca" - ExperimentRunner.py:105-117,
                # kept; `arithmetic_mode` is this package's: which summation order produced the rows of each dimension)
                provenance = {"arithmetic_mode": ",".join(f"d{dim}={self.arithmetic_modes[dim]}" for dim in self.dimensions)}
                logger.set_experiment_attributes({
                    "budget_factor": f"{self.budget_factor}",
                    "doe_factor": f"{self.doe_factor}",
                    "acquisition_function": f"{self.acquisition_function}",
                    **provenance
                })
                if algorithm == "pca":
                    logger.set_experiment_attributes({
                        "pca_components": f"{self.pca_components}",
                        "var_threshold": f"{self.var_threshold}",
                        **provenance
                    })
                optimizer_class = Vanilla_BO if algorithm == "vanilla" else PCA_BO
                for time_profile in getattr(optimizer_class, "TIME_PROFILES", []):
                    logger.add_run_attribute(f"{time_profile}_time", 0.0)
                logger.add_run_attribute("time", 0.0)

                if self.batched > 1 and HAVE_IOH:     # pragma: no cover
                    import warnings
                    warnings.warn("batched > 1 is ignored while `ioh` is installed: the lock-step driver evaluates the in-repo "
                                  "BBOB problems (pcabo.bbob) and replays a run's rows into the logger afterwards; the runs "
                                  "go one at a time through ioh's own suite and Analyzer instead.", RuntimeWarning)
                if self.batched > 1 and not HAVE_IOH:
                    self._run_batched(logger, ebar, algorithm)
                    logger.close()
                    continue
                for problem in self._problems(logger):
                    dim = problem.meta_data.n_variables
                    problem_id = problem.meta_data.problem_id
                    instance = problem.meta_data.instance
                    maximization = bool(problem.meta_data.optimization_type.value)
                    run_num = self.instances.index(instance)

                    budget = self.budget_factor * dim + 50
                    n_doe = int(self.doe_factor * dim)
                    random_seed = 1000 * problem_id + 10 * dim + instance

                    with bar(total=budget, position=1, desc="", leave=False) as pbar:
                        pbar.set_description(f"{algorithm} | {dim}-dim | F-{problem_id} | run-{run_num + 1}")
                        if self.verbose:
                            pbar.write(f"\nRunning {algorithm} | {dim}-dim | F-{problem_id} | run-{run_num + 1}:\n")
                        common = dict(budget=budget, n_DoE=n_doe, acquisition_function=self.acquisition_function,
                                      random_seed=random_seed, maximization=maximization, verbose=self.verbose,
                                      DoE_parameters=self.doe_params, pbar=pbar if self.progress else None,
                                      device=self.device)
                        if algorithm == "vanilla":
                            optimizer = Vanilla_BO(**common)
                        else:
                            optimizer = PCA_BO(var_threshold=self.var_threshold, **common)

                        start_time = time()
                        optimizer(problem=problem)
                        elapsed = time() - start_time
                        logger.set_run_attribute("time", elapsed)
                        for time_profile, total_profile_time in optimizer.total_times.items():
                            logger.set_run_attribute(f"{time_profile}_time", total_profile_time)
                        self.results.append({"algorithm": algorithm, "problem_id": problem_id, "dim": dim,
                                             "instance": instance, "best": optimizer.current_best, "time": elapsed,
                                             "iterations": budget - n_doe, **optimizer.total_times})
                        if self.verbose:
                            pbar.write(f"The distance from optimum is: "
                                       f"{norm(problem.state.current_best.x - problem.optimum.x)}")
                            pbar.write(f"The regret is: {problem.state.current_best.y - problem.optimum.y}")
                        pbar.close()
                        ebar.update(1)
                logger.close()
