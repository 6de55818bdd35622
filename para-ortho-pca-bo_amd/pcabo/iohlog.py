"""IOHprofiler-format run logger (the data format on the far side of the hot path).

The reference logs through `ioh.iohcpp.logger.Analyzer` (ExperimentRunner.py:94-126,186-190,199-200) with
triggers=[ALWAYS], additional_properties=[RAWYBEST], store_positions=True, and plot_results.py reads those files
back.  `ioh` is a third-party C++ extension that is absent here, so this module writes the same on-disk layout
(IOHprofiler 0.3.18, the version of the files the reference commits under {pca,vanilla}-experiment/):

    <root>/<folder>/IOHprofiler_f<id>_<name>.json                 one per function: meta data + one entry per run
    <root>/<folder>/data_f<id>_<name>/IOHprofiler_f<id>_DIM<d>.dat   one block per run:
        evaluations raw_y raw_y_best x0 ... x{d-1}                header line
        <eval> <raw_y %.10f> <best so far %.10f> <x %.6f> ...

`raw_y` is the objective without the instance's f_opt shift, as in the reference's files.  The layout facts are
pinned by tests/golden/ref_kats_dim5.json ("dat_header", "dat_first_row", "json_*").
When `ioh` is importable the ExperimentRunner uses the real Analyzer instead of this module.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional

import numpy as np

IOH_VERSION = "0.3.18"


def _num(v) -> str:
    """Numbers the way the 0.3.18 writer prints them in the .json (shortest round-trip repr, integers bare)."""
    v = float(v)
    return repr(int(v)) if v == int(v) and abs(v) < 1e15 else repr(v)


class _Run:
    def __init__(self, instance: int, dim: int, maximization: bool):
        self.instance, self.dim, self.maximization = instance, dim, maximization
        self.evals = 0
        self.best_y = -np.inf if maximization else np.inf
        self.best_x: Optional[np.ndarray] = None
        self.best_evals = 0
        self.attributes: Dict[str, float] = {}


class Analyzer:
    """Subset of the ioh Analyzer used by the reference's runner: experiment / run attributes, attach by wrapping
    the problem, one .dat block per run, .json written on close()."""

    def __init__(self, root: str, folder_name: str, algorithm_name: str, algorithm_info: str = "",
                 store_positions: bool = True, suite: str = "BBOB"):
        base = os.path.join(root, folder_name)
        path, i = base, 0
        while os.path.exists(path):        # the ioh logger never overwrites: folder, folder-1, folder-2, ...
            i += 1
            path = f"{base}-{i}"
        os.makedirs(path)
        self.output_directory = path
        self.algorithm_name, self.algorithm_info = algorithm_name, algorithm_info
        self.store_positions = store_positions
        self.suite = suite
        self.experiment_attributes: Dict[str, str] = {}
        self.run_attribute_names: List[str] = []
        self._functions: Dict[int, dict] = {}        # fid -> {"name", "maximization", "scenarios": {dim: [runs]}}
        self._run: Optional[_Run] = None
        self._dat = None
        self._pending: Dict[str, float] = {}

    # ---- attributes (ExperimentRunner.py:105-126,186-190) -------------------------------------------------------
    def set_experiment_attributes(self, attrs: Dict[str, str]) -> None:
        self.experiment_attributes = dict(attrs)      # a second call replaces the first (see the committed .json files)

    def add_run_attribute(self, name: str, value: float = 0.0) -> None:
        if name not in self.run_attribute_names:
            self.run_attribute_names.append(name)
        self._pending[name] = float(value)

    def set_run_attribute(self, name: str, value: float) -> None:
        if name not in self.run_attribute_names:
            raise KeyError(f"run attribute {name!r} was not added before the run")
        (self._run.attributes if self._run is not None else self._pending)[name] = float(value)

    # ---- per-run life cycle ---------------------------------------------------------------------------------------
    def start_run(self, function_id: int, function_name: str, instance: int, dim: int, maximization: bool) -> None:
        self.end_run()
        fn = self._functions.setdefault(function_id, {"name": function_name, "maximization": maximization, "scenarios": {}})
        ddir = os.path.join(self.output_directory, f"data_f{function_id}_{function_name}")
        os.makedirs(ddir, exist_ok=True)
        rel = f"data_f{function_id}_{function_name}/IOHprofiler_f{function_id}_DIM{dim}.dat"
        fn["scenarios"].setdefault(dim, {"path": rel, "runs": []})
        self._dat = open(os.path.join(self.output_directory, rel), "a")
        cols = "evaluations raw_y raw_y_best"
        if self.store_positions:
            cols += "".join(f" x{i}" for i in range(dim))
        self._dat.write(cols + "\n")
        self._run = _Run(instance, dim, maximization)
        self._run.attributes = {k: self._pending.get(k, 0.0) for k in self.run_attribute_names}
        self._run_fid = function_id

    def log(self, x: np.ndarray, raw_y: float) -> None:
        r = self._run
        r.evals += 1
        better = raw_y > r.best_y if r.maximization else raw_y < r.best_y
        if better:
            r.best_y, r.best_x, r.best_evals = float(raw_y), np.array(x, dtype=np.float64), r.evals
        line = f"{r.evals} {raw_y:.10f} {r.best_y:.10f}"
        if self.store_positions:
            line += "".join(f" {v:.6f}" for v in np.asarray(x, dtype=np.float64).ravel())
        self._dat.write(line + "\n")

    def end_run(self) -> None:
        if self._run is None:
            return
        r = self._run
        self._dat.close()
        self._dat = None
        if r.evals:
            self._functions[self._run_fid]["scenarios"][r.dim]["runs"].append(r)
        self._run = None

    def close(self) -> None:
        self.end_run()
        for fid, fn in self._functions.items():
            with open(os.path.join(self.output_directory, f"IOHprofiler_f{fid}_{fn['name']}.json"), "w") as fh:
                fh.write(self._render(fid, fn))

    # ---- .json in the 0.3.18 layout (tab indented, one run per line) ------------------------------------------------
    def _render(self, fid: int, fn: dict) -> str:
        q = json.dumps
        exp = ", ".join("{" + f"{q(k)}: {q(str(v))}" + "}" for k, v in sorted(self.experiment_attributes.items()))
        attrs = ["evaluations", "raw_y", "raw_y_best"]
        out = ["{", f'\t"version": {q(IOH_VERSION)}, ', f'\t"suite": {q(self.suite)}, ', f'\t"function_id": {fid}, ',
               f'\t"function_name": {q(fn["name"])}, ', f'\t"maximization": {q(bool(fn["maximization"]))}, ',
               f'\t"algorithm": {{"name": {q(self.algorithm_name)}, "info": {q(self.algorithm_info)}}},',
               f'\t"experiment_attributes": [{exp}],',
               f'\t"run_attributes": [{", ".join(q(n) for n in sorted(self.run_attribute_names))}],',
               f'\t"attributes": [{", ".join(q(a) for a in attrs)}],', '\t"scenarios": [']
        scen = []
        for dim in sorted(fn["scenarios"]):
            sc = fn["scenarios"][dim]
            runs = []
            for r in sc["runs"]:
                bx = ", ".join(_num(v) for v in r.best_x)
                ra = "".join(f', {q(n)}: {_num(r.attributes.get(n, 0.0)) if r.attributes.get(n, 0.0) == 0 else repr(float(r.attributes[n]))}'
                             for n in sorted(self.run_attribute_names))
                runs.append(f'\t\t\t{{"instance": {r.instance}, "evals": {r.evals}, "best": {{"evals": {r.best_evals}, '
                            f'"y": {_num(r.best_y)}, "x": [{bx}]}}{ra}}}')
            scen.append(f'\t\t{{"dimension": {dim},\n\t\t"path": {q(sc["path"])},\n\t\t"runs": [\n' + ",\n".join(runs) + "\n\t\t]\n\t\t}")
        out.append(",\n".join(scen))
        out += ["\t]", "}"]
        return "\n".join(out) + "\n"


class LoggedProblem:
    """What `suite.attach_logger(logger)` does to an ioh problem, for a duck-typed problem: every call is logged.
    Forwards the attributes the optimisers read (`meta_data`, `bounds`, `optimum`)."""

    def __init__(self, problem, logger: Analyzer):
        self._problem, self._logger = problem, logger
        md = problem.meta_data
        self.meta_data, self.bounds = md, problem.bounds
        self.optimum = getattr(problem, "optimum", None)
        self._f_opt = float(getattr(problem, "f_opt", 0.0))
        logger.start_run(md.problem_id, md.name, md.instance, md.n_variables, bool(md.optimization_type.value))
        self.state = _State(self)

    def __call__(self, x):
        y = self._problem(x)
        self._logger.log(np.asarray(x, dtype=np.float64).ravel(), y - self._f_opt)
        return y


class _State:
    """`problem.state.current_best.{x,y}` as read by the runner's verbose report (ExperimentRunner.py:193-195)."""

    def __init__(self, lp: LoggedProblem):
        self._lp = lp

    @property
    def current_best(self):
        from types import SimpleNamespace
        r = self._lp._logger._run
        return SimpleNamespace(x=r.best_x, y=r.best_y + self._lp._f_opt)


def read_dat(path: str):
    """Blocks of a .dat file -> list of float arrays (rows x columns); the reader used by the tests and the bench."""
    runs, cur = [], None
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "evaluations":
                cur = []
                runs.append(cur)
            else:
                cur.append([float(v) for v in t])
    return [np.array(r) for r in runs]
