"""Host-side BBOB objectives used as synthetic workloads (problem side of the boundary).

f15 (Rastrigin rotated, the function BASELINE.json names) and f20 (Schwefel, the second function of the
reference's `--quick` configuration, main.py:103-109) are restated and PINNED by the reference's own logged
runs (tests/golden/ref_kats_dim5.json).

f16-f19 and f21-f24 (BASELINE.json configs[2] / configs[3]: "f15/f16/f17", "f15-f24") are restated from the published
COCO legacy definitions (Hansen, Finck, Ros, Auger 2009, "Real-parameter black-box optimization benchmarking 2009:
noiseless functions definitions", and the bbob-legacy code path that `ioh` 0.3.18 follows: seeds, rotation and x_opt
generators as for f15).  The reference holds NO known answers for them (only f15 and f20 runs are committed), so they
are **parity unpinned**: checked here against their defining properties only (optimum value at x_opt, penalty outside
the box, invariances; tests/test_bbob_functions.py).

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


def _rseed(function_id: int, instance: int) -> int:
    """COCO legacy: f4 shares f3's seed and f18 shares f17's (suite_bbob.c `rseed_3`, `rseed_17`)."""
    base = {4: 3, 18: 17}.get(function_id, function_id)
    return base + 10000 * instance


def _xopt(dim: int, rseed: int) -> np.ndarray:
    """COCO `bbob2009_compute_xopt`."""
    v = bbob_uniform(dim, rseed)
    xopt = 8.0 * np.floor(1e4 * v) / 1e4 - 4.0
    xopt[xopt == 0.0] = -1e-5
    return xopt


def _boundary_penalty(x: np.ndarray) -> float:
    out = np.abs(x) - 5.0
    return float(np.sum(np.where(out > 0.0, out * out, 0.0)))


def _fopt(function_id: int, instance: int) -> float:
    """COCO `bbob2009_compute_fopt`.  NOT verifiable from the reference's files."""
    rseed = _rseed(function_id, instance)
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




# ---- f16-f19, f21-f24: restated from the published definitions, parity unpinned (module docstring) -------------------
class _RotatedState:
    """x_opt, R = rot(rseed + 10^6), Q = rot(rseed) and M = R diag(c_k) Q as the legacy code builds them."""

    def __init__(self, dim: int, instance: int, function_id: int, cond_base: float = 1.0):
        rseed = _rseed(function_id, instance)
        self.dim, self.rseed = dim, rseed
        self.xopt = _xopt(dim, rseed)
        self.rot_r = bbob_rotation(dim, rseed + 1000000)
        self.rot_q = bbob_rotation(dim, rseed)
        self.scale = cond_base ** (np.arange(dim) / (dim - 1.0))          # c_k = cond_base^(k/(D-1))
        self.m = self.rot_r @ (self.scale[:, None] * self.rot_q)


class _F16State(_RotatedState):
    """Weierstrass: M = R Lambda^(1/100) Q, i.e. c_k = (1/sqrt(100))^(k/(D-1))."""

    def __init__(self, dim, instance):
        super().__init__(dim, instance, 16, 1.0 / math.sqrt(100.0))
        self.ak = 0.5 ** np.arange(12)
        self.bk = 3.0 ** np.arange(12)
        self.f0 = float(np.sum(self.ak * np.cos(2.0 * math.pi * self.bk * 0.5)))


def f16_raw(x, st: _F16State) -> float:
    x = np.asarray(x, dtype=np.float64)
    z = st.m @ _t_osz(st.rot_r @ (x - st.xopt))
    s = float(np.sum(np.cos(2.0 * math.pi * np.outer(z + 0.5, st.bk)) * st.ak))
    return 10.0 * (s / st.dim - st.f0) ** 3 + (10.0 / st.dim) * _boundary_penalty(x)


class _F17State(_RotatedState):
    """Schaffers F7: z = Lambda^cond Q T_asy^0.5(R (x - x_opt)); conditioning 10 (f17) or 1000 (f18, f17's seed)."""
    COND = 10.0
    FID = 17

    def __init__(self, dim, instance):
        super().__init__(dim, instance, self.FID, math.sqrt(self.COND))
        self.m = self.scale[:, None] * self.rot_q                           # rows of Q scaled: no second rotation


class _F18State(_F17State):
    COND = 1000.0
    FID = 18


def f17_raw(x, st: _F17State) -> float:
    x = np.asarray(x, dtype=np.float64)
    z = st.m @ _t_asy(st.rot_r @ (x - st.xopt), 0.5)
    t = z[:-1] ** 2 + z[1:] ** 2
    s = float(np.sum(t ** 0.25 * (1.0 + np.sin(50.0 * t ** 0.1) ** 2)))
    return (s / (st.dim - 1.0)) ** 2 + 10.0 * _boundary_penalty(x)


class _F19State:
    """Composite Griewank-Rosenbrock F8F2: z = max(1, sqrt(D)/8) R x + 0.5 with R = rot(rseed); no x_opt shift."""

    def __init__(self, dim, instance):
        rseed = _rseed(19, instance)
        self.dim = dim
        self.m = max(1.0, math.sqrt(dim) / 8.0) * bbob_rotation(dim, rseed)
        # the optimum z = 1 maps back to x_opt = M^-1 (1 - 0.5)
        self.xopt = np.linalg.solve(self.m, np.full(dim, 0.5))


def f19_raw(x, st: _F19State) -> float:
    z = st.m @ np.asarray(x, dtype=np.float64) + 0.5
    c1 = z[:-1] ** 2 - z[1:]
    c2 = 1.0 - z[:-1]
    t = 100.0 * c1 * c1 + c2 * c2
    return 10.0 + 10.0 * float(np.sum(t / 4000.0 - np.cos(t))) / (st.dim - 1.0)


class _GallagherState:
    """Gallagher's Gaussian peaks (101 for f21, 21 for f22): peak heights 10, 1.1 .. 9.1; the global peak's condition is
    sqrt(1000) (f21) / 1000 (f22), the others 1000^(j/(P-2)) in a seeded random order; every peak has its own seeded
    permutation of the axis scales; peak positions R-rotated uniform draws in [-c, b - c], the global one shrunk by 0.8."""

    def __init__(self, dim, instance, function_id, peaks, b, c, first_cond):
        rseed = _rseed(function_id, instance)
        self.dim, self.peaks = dim, peaks
        self.rot = bbob_rotation(dim, rseed)
        order = np.argsort(bbob_uniform(peaks - 1, rseed), kind="stable")
        conds = np.empty(peaks)
        conds[0] = first_cond
        conds[1:] = 1000.0 ** (order / (peaks - 2.0))
        self.heights = np.empty(peaks)
        self.heights[0] = 10.0
        self.heights[1:] = np.arange(peaks - 1) / (peaks - 2.0) * 8.0 + 1.1
        self.scales = np.empty((peaks, dim))
        for i in range(peaks):
            perm = np.argsort(bbob_uniform(dim, rseed + 1000 * i), kind="stable")
            self.scales[i] = conds[i] ** (perm / (dim - 1.0) - 0.5)
        u = bbob_uniform(dim * peaks, rseed).reshape(peaks, dim)
        self.xopt = 0.8 * (b * u[0] - c)
        self.centres = (b * u - c) @ self.rot.T                             # x_local[:, j] = R (b u_j - c)
        self.centres[0] *= 0.8


class _F21State(_GallagherState):
    def __init__(self, dim, instance):
        super().__init__(dim, instance, 21, 101, 10.0, 5.0, math.sqrt(1000.0))


class _F22State(_GallagherState):
    def __init__(self, dim, instance):
        super().__init__(dim, instance, 22, 21, 9.8, 4.9, 1000.0)


def _osz_scalar(f: float) -> float:
    if f > 0:
        t = math.log(f) / 0.1
        return math.exp(t + 0.49 * (math.sin(t) + math.sin(0.79 * t))) ** 0.1
    if f < 0:
        t = math.log(-f) / 0.1
        return -(math.exp(t + 0.49 * (math.sin(0.55 * t) + math.sin(0.31 * t))) ** 0.1)
    return 0.0


def gallagher_raw(x, st: _GallagherState) -> float:
    x = np.asarray(x, dtype=np.float64)
    tx = st.rot @ x
    diff = tx[None, :] - st.centres
    expo = (-0.5 / st.dim) * np.sum(st.scales * diff * diff, axis=1)
    f = 10.0 - float(np.max(st.heights * np.exp(expo)))
    f = _osz_scalar(f)
    return f * f + _boundary_penalty(x)


class _F23State(_RotatedState):
    """Katsuura: z = R Lambda^100 Q (x - x_opt) with M = rot(rseed + 10^6) diag(sqrt(100)^(k/(D-1))) rot(rseed)."""

    def __init__(self, dim, instance):
        super().__init__(dim, instance, 23, math.sqrt(100.0))
        self.pow2 = 2.0 ** np.arange(1, 33)


def f23_raw(x, st: _F23State) -> float:
    x = np.asarray(x, dtype=np.float64)
    z = st.m @ (x - st.xopt)
    t = np.outer(z, st.pow2)
    inner = np.sum(np.abs(t - np.floor(t + 0.5)) / st.pow2, axis=1)
    terms = (1.0 + np.arange(1, st.dim + 1) * inner) ** (10.0 / st.dim ** 1.2)
    return 10.0 / st.dim / st.dim * (float(np.prod(terms)) - 1.0) + _boundary_penalty(x)


class _F24State:
    """Lunacek bi-Rastrigin: x_opt = +-mu0/2 with the signs of a seeded gaussian draw."""

    def __init__(self, dim, instance):
        rseed = _rseed(24, instance)
        self.dim = dim
        self.mu0, self.d = 2.5, 1.0
        self.s = 1.0 - 0.5 / (math.sqrt(dim + 20.0) - 4.1)
        self.mu1 = -math.sqrt((self.mu0 ** 2 - self.d) / self.s)
        g = bbob_gauss(dim, rseed)
        self.xopt = np.where(g < 0.0, -0.5 * self.mu0, 0.5 * self.mu0)
        self.rot_r = bbob_rotation(dim, rseed + 1000000)
        self.rot_q = bbob_rotation(dim, rseed)
        self.cond = math.sqrt(100.0) ** (np.arange(dim) / (dim - 1.0))


def f24_raw(x, st: _F24State) -> float:
    x = np.asarray(x, dtype=np.float64)
    xh = 2.0 * np.where(st.xopt < 0.0, -x, x)
    z = st.rot_r @ (st.cond * (st.rot_q @ (xh - st.mu0)))
    s1 = float(np.sum((xh - st.mu0) ** 2))
    s2 = float(np.sum((xh - st.mu1) ** 2))
    s3 = float(np.sum(np.cos(2.0 * math.pi * z)))
    return min(s1, st.d * st.dim + st.s * s2) + 10.0 * (st.dim - s3) + 1e4 * _boundary_penalty(x)


FUNCTIONS = {15: ("RastriginRotated", _F15State, f15_raw), 16: ("Weierstrass", _F16State, f16_raw),
             17: ("Schaffers10", _F17State, f17_raw), 18: ("Schaffers1000", _F18State, f17_raw),
             19: ("GriewankRosenBrock", _F19State, f19_raw), 20: ("Schwefel", _F20State, f20_raw),
             21: ("Gallagher101", _F21State, gallagher_raw), 22: ("Gallagher21", _F22State, gallagher_raw),
             23: ("Katsuura", _F23State, f23_raw), 24: ("LunacekBiRastrigin", _F24State, f24_raw)}
PINNED_BY_REFERENCE_DATA = (15, 20)      # every other function id here is parity unpinned

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
