"""Host-side BBOB objectives used as synthetic workloads (problem side of the boundary).

f15 (Rastrigin rotated, the function BASELINE.json names) and f20 (Schwefel, the second function of the
reference's `--quick` configuration, main.py:103-109) are restated; both are pinned by the reference's own logged
runs (tests/golden/ref_kats_dim5.json).

The reference obtains its objectives from the third-party `ioh` package
(reference: Algorithms/Experiment/ExperimentRunner.py:90, example.py:82-86), which is
not installed here.  This module restates the COCO/IOH *legacy* definitions so that `PCA_BO` /
`Vanilla_BO` can be driven without `ioh`; the known answers are the rows the reference itself ships in
{pca,vanilla}-experiment/data_f*/IOHprofiler_f*_DIM5.dat and the best points of the matching .json files.

The objective is *outside* the accelerated path: it runs on the
host, one point at a time, exactly like `problem(x)` does in the reference
(PCA_BO.py:263, AbstractBayesianOptimizer.py:163).

`BBOBProblem` duck-types the three attributes of an ioh `RealSingleObjective` that the
optimizer reads (AbstractAlgorithm.py:73-83,246-268): `meta_data.n_variables`,
`meta_data.optimization_type`, `bounds.lb/.ub`.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np

__all__ = ["BBOBProblem", "bbob_uniform", "bbob_gauss", "bbob_rotation", "f15_raw", "f20_raw", "get_problem", "FUNCTIONS"]


def bbob_uniform(n: int, seed: int) -> np.ndarray:
    """Legacy BBOB uniform generator (Park-Miller with a 32 entry Bays-Durham shuffle)."""
    seed = abs(int(seed))
    if seed < 1:
        seed = 1
    a = seed
    table = [0] * 32
    for i in range(39, -1, -1):
        t = a // 127773
        a = 16807 * (a - t * 127773) - 2836 * t
        if a < 0:
            a += 2147483647
        if i < 32:
            table[i] = a
    r = table[0]
    out = np.empty(n)
    for i in range(n):
        t = a // 127773
        a = 16807 * (a - t * 127773) - 2836 * t
        if a < 0:
            a += 2147483647
        j = r // 67108865
        r = table[j]
        table[j] = a
        v = r / 2.147483647e9
        out[i] = v if v != 0.0 else 1e-99
    return out


def bbob_gauss(n: int, seed: int) -> np.ndarray:
    u = bbob_uniform(2 * n, seed)
    g = np.sqrt(-2.0 * np.log(u[:n])) * np.cos(2.0 * math.pi * u[n:])
    g[g == 0.0] = 1e-99
    return g


def bbob_rotation(dim: int, seed: int) -> np.ndarray:
    """Orthogonal matrix: Gram-Schmidt over the columns of a column-major gaussian fill."""
    b = bbob_gauss(dim * dim, seed).reshape(dim, dim).T.copy()  # B[i][j] = g[j*dim + i]
    for i in range(dim):
        for j in range(i):
            b[:, i] -= np.dot(b[:, i], b[:, j]) * b[:, j]
        b[:, i] /= math.sqrt(np.dot(b[:, i], b[:, i]))
    return b


def _t_osz(x: np.ndarray) -> np.ndarray:
    out = np.zeros_like(x)
    pos, neg = x > 0, x < 0
    t = np.log(x[pos]) / 0.1
    out[pos] = np.exp(t + 0.49 * (np.sin(t) + np.sin(0.79 * t))) ** 0.1
    t = np.log(-x[neg]) / 0.1
    out[neg] = -(np.exp(t + 0.49 * (np.sin(0.55 * t) + np.sin(0.31 * t))) ** 0.1)
    return out


def _t_asy(x: np.ndarray, beta: float) -> np.ndarray:
    dim = x.size
    out = x.copy()
    pos = x > 0
    idx = np.arange(dim)[pos]
    out[pos] = x[pos] ** (1.0 + beta * (idx / (dim - 1.0)) * np.sqrt(x[pos]))
    return out


def _fopt(function_id: int, instance: int) -> float:
    """COCO `bbob2009_compute_fopt`.  NOT verifiable from the reference's files."""
    rseed = function_id + 10000 * instance
    g1 = bbob_gauss(1, rseed)[0]
    g2 = bbob_gauss(1, rseed + 1)[0]
    v = 100.0 * 100.0 * g1 / g2
    v = math.floor(abs(v) + 0.5) * (1.0 if v >= 0 else -1.0) / 100.0
    return min(1000.0, max(-1000.0, v))


class _F15State:
    def __init__(self, dim: int, instance: int):
        rseed = 15 + 10000 * instance
        v = bbob_uniform(dim, rseed)
        xopt = 8.0 * np.floor(1e4 * v) / 1e4 - 4.0
        xopt[xopt == 0.0] = -1e-5
        self.xopt = xopt
        self.rot_r = bbob_rotation(dim, rseed + 1000000)
        rot_q = bbob_rotation(dim, rseed)
        lam = np.sqrt(10.0) ** (np.arange(dim) / (dim - 1.0))
        self.m = self.rot_r @ (lam[:, None] * rot_q)
        self.dim = dim


def f15_raw(x: np.ndarray, state: _F15State) -> float:
    """Rastrigin rotated, value before the f_opt shift (the `raw_y` column of the .dat files)."""
    y = state.rot_r @ (np.asarray(x, dtype=np.float64) - state.xopt)
    z = state.m @ _t_asy(_t_osz(y), 0.2)
    return float(10.0 * (state.dim - np.sum(np.cos(2.0 * math.pi * z))) + np.dot(z, z))


class _F20State:
    """BBOB f20 (Schwefel): x_opt = +-4.2096874637/2 with the signs of a seeded uniform draw; the same draw decides
    the sign flip of x; conditioning 10^(i / (2 (D-1)))."""

    def __init__(self, dim: int, instance: int):
        rseed = 20 + 10000 * instance
        u = bbob_uniform(dim, rseed)
        self.sign = np.where(u < 0.5, -1.0, 1.0)
        self.xopt = self.sign * 0.5 * 4.2096874637
        self.offset = 2.0 * np.abs(self.xopt)
        self.cond = np.sqrt(10.0) ** (np.arange(dim) / (dim - 1.0))
        self.dim = dim


def f20_raw(x: np.ndarray, state: _F20State) -> float:
    """Schwefel x sin(sqrt|x|) with the BBOB variable transformations, value before the f_opt shift."""
    xh = 2.0 * state.sign * np.asarray(x, dtype=np.float64)
    zh = xh.copy()
    zh[1:] += 0.25 * (xh[:-1] - state.offset[:-1])
    z = 100.0 * (state.cond * (zh - state.offset) + state.offset)
    out = np.abs(z) - 500.0
    penalty = float(np.sum(np.where(out > 0.0, out * out, 0.0)))
    total = float(np.sum(z * np.sin(np.sqrt(np.abs(z)))))
    return 0.01 * (penalty + 418.9828872724339 - total / state.dim)


FUNCTIONS = {15: ("RastriginRotated", _F15State, f15_raw), 20: ("Schwefel", _F20State, f20_raw)}

_MIN = SimpleNamespace(value=0, name="MIN")


class BBOBProblem:
    """Minimal ioh-like single-objective problem (minimisation, box [-5,5]^d)."""

    def __init__(self, function_id: int, instance: int, dimension: int, add_fopt: bool = True):
        if function_id not in FUNCTIONS:
            raise NotImplementedError(f"BBOB f{function_id} is not restated here (available: {sorted(FUNCTIONS)}); "
                                      "pass an ioh problem instead")
        name, state_cls, self._raw = FUNCTIONS[function_id]
        if dimension < 2:
            raise ValueError("BBOB problems need dimension >= 2")
        self._state = state_cls(dimension, instance)
        self.f_opt = _fopt(function_id, instance) if add_fopt else 0.0
        self.meta_data = SimpleNamespace(
            n_variables=int(dimension), problem_id=int(function_id), instance=int(instance),
            name=name, optimization_type=_MIN)
        self.bounds = SimpleNamespace(lb=np.full(dimension, -5.0), ub=np.full(dimension, 5.0))
        self.optimum = SimpleNamespace(x=self._state.xopt.copy(), y=self.f_opt)
        self.evaluations = 0
        self.log = []            # (raw_y, x) of every evaluated point, like the Analyzer rows
        self.best_raw = math.inf

    def raw(self, x) -> float:
        return self._raw(np.asarray(x, dtype=np.float64).ravel(), self._state)

    def __call__(self, x) -> float:
        x = np.asarray(x, dtype=np.float64).ravel()
        r = self._raw(x, self._state)
        self.evaluations += 1
        self.best_raw = min(self.best_raw, r)
        self.log.append((r, x.copy()))
        return r + self.f_opt


def get_problem(function_id: int, instance: int, dimension: int, **kw) -> BBOBProblem:
    """Same argument order as `ioh.get_problem` (reference: example.py:82-86)."""
    return BBOBProblem(function_id, instance, dimension, **kw)
