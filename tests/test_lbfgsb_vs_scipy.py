"""This library's host L-BFGS-B (csrc/lbfgsb.cpp) pinned against scipy's own implementation - the
third-party code the reference reaches through botorch (PCA_BO.py:607-614).  Host-only: no GPU."""
import numpy as np
import pytest
import torch
from scipy.optimize import minimize

import pcabo_oracle as O
from pcabo.bbob import BBOBProblem


def rosen(x):
    f = np.sum(100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
    g = np.zeros_like(x)
    g[:-1] = -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
    g[1:] += 200 * (x[1:] - x[:-1] ** 2)
    return f, g


def _both(native, fun, x0, bounds, maxiter):
    ref = minimize(fun, x0, jac=True, method="L-BFGS-B", bounds=bounds, options={"maxiter": maxiter})
    mine = native.lbfgsb_minimize(fun, x0, bounds, maxiter=maxiter)
    return ref, mine


def test_bounded_rosenbrock_same_path_as_scipy(native):
    rng = np.random.default_rng(1)
    for _ in range(25):
        n = int(rng.integers(2, 30))
        x0 = rng.uniform(-2, 2, n)
        lo = rng.uniform(-3, 0.5, n)
        hi = lo + rng.uniform(0.5, 4, n)
        ref, mine = _both(native, rosen, x0, list(zip(lo, hi)), int(rng.integers(5, 300)))
        assert (ref.nit, ref.nfev) == (mine["nit"], mine["nfev"])
        assert np.abs(ref.x - mine["x"]).max() < 1e-8
        assert (0 if ref.success else (1 if ref.status == 1 else 2)) == mine["warnflag"]


def test_unbounded_and_half_bounded_variables(native):
    rng = np.random.default_rng(5)
    for _ in range(10):
        n = 8
        x0 = rng.uniform(-2, 2, n)
        bounds = [(None, None), (0.0, None), (None, 1.5), (-1.0, 1.0)] * 2
        ref, mine = _both(native, rosen, x0, bounds, 200)
        assert (ref.nit, ref.nfev) == (mine["nit"], mine["nfev"])
        assert np.abs(ref.x - mine["x"]).max() < 1e-8


def test_joint_five_restart_acquisition_problem_same_path_as_scipy(native):
    torch.set_num_threads(1)
    p = BBOBProblem(15, 0, 10)
    o = O.OraclePCABO(budget=150, n_DoE=30, random_seed=15100, record=True)
    o(p, 10, np.array([-5.0, 5.0]), max_iters=2)
    for rec in o.records:
        gp = O.ExactGP(rec.wpca.Z, rec.f, rec.norm_bounds)
        acq = O.Acquisition(gp, rec.best_f, False)
        b, k = 5, rec.wpca.k
        lo, hi = np.tile(rec.acq_bounds[0], b), np.tile(rec.acq_bounds[1], b)

        def fun(x):
            v, g = acq.value_and_grad(x.reshape(b, k))
            return -float(v.sum()), -g.reshape(-1)

        for s in (0, 5):
            x0 = np.clip(rec.trace.ics[s:s + 5].reshape(-1), lo, hi)
            ref, mine = _both(native, fun, x0, list(zip(lo, hi)), 200)
            assert (ref.nit, ref.nfev) == (mine["nit"], mine["nfev"])
            assert np.abs(ref.x - mine["x"]).max() < 1e-9


def test_abnormal_line_search_and_memoised_repeats_like_scipy(native):
    """Inconsistent objective (value rises along the descent direction): the line search shrinks until the trial
    point repeats; scipy's ScalarFunction memoises such repeats (not called, not counted) and the run ends as
    ABNORMAL (warnflag 2) - the case in which botorch redraws initial conditions."""
    for n, bounds in ((1, [(-10, 10)]), (3, [(-10, 10)] * 3), (3, [(None, None)] * 3)):
        calls = {"scipy": 0, "mine": 0}

        def make(tag):
            def fun(x):
                calls[tag] += 1
                return float(np.sum(x)), -np.ones_like(x)
            return fun

        ref = minimize(make("scipy"), np.full(n, 0.5), jac=True, method="L-BFGS-B", bounds=bounds, options={"maxiter": 50})
        mine = native.lbfgsb_minimize(make("mine"), np.full(n, 0.5), bounds, maxiter=50)
        assert ref.status == 2 and mine["warnflag"] == 2 and mine["task"] == 70
        assert (ref.nit, ref.nfev) == (mine["nit"], mine["nfev"])
        assert calls["scipy"] == calls["mine"]
        assert np.array_equal(ref.x, mine["x"])


def test_vector_kernels_take_the_scalar_iterates(native):
    """The AVX2 kernels of lbfgsb.cpp (row-major mirror reductions, per-variable chains) must not change a single
    bit of any iterate: the same bounded problems with the vector kernels and with the scalar loops
    (pcabo_lbfgsb_set_vector_kernels)."""
    def run_all():
        out = []
        for seed in range(6):
            rng = np.random.default_rng(seed)
            nv = [7, 40, 85, 165, 33, 120][seed]
            c = rng.uniform(-0.2, 1.2, nv)
            s = rng.uniform(0.5, 2.0, nv)

            def fun(x):
                d = x - c
                f = float(np.sum(s * d * d + 0.1 * d ** 4) + 0.05 * np.sum(np.cos(5 * x + np.roll(x, -1))))
                g = 2 * s * d + 0.4 * d ** 3 - 0.25 * np.sin(5 * x + np.roll(x, -1)) - 0.05 * np.roll(np.sin(5 * x + np.roll(x, -1)), 1)
                return f, g
            r = native.lbfgsb_minimize(fun, rng.uniform(0, 1, nv), [(0.0, 1.0)] * nv, maxiter=200)
            out.append((r["nit"], r["nfev"], r["task"], float(r["fun"]).hex(), [float(v).hex() for v in r["x"]]))
        return out

    was = native.lbfgsb_set_vector_kernels(False)
    try:
        scalar = run_all()
    finally:
        native.lbfgsb_set_vector_kernels(was)
    vector = run_all()
    assert scalar == vector
    assert all(r[0] > 3 for r in vector)            # real optimisation runs, not immediate exits


def test_tree_order_of_the_device_optimiser_keeps_scipys_counts(native):
    """The summation order in which the device-resident optimiser steps (64-lane tree over the variables, reciprocal pivots in the
    small triangular solves; Lbfgsb::set_sum_order(1), the device's bit-for-bit twin - tests/test_gpu_device_lbfgsb.py) on the same
    35 bounded problems and the joint 5-restart acquisition problem: iteration and evaluation counts are scipy's on every one of
    them, end points within the same tolerance (the gate for a second arithmetic order)."""
    was = native.lbfgsb_set_sum_order(1)
    try:
        test_bounded_rosenbrock_same_path_as_scipy(native)
        test_unbounded_and_half_bounded_variables(native)
        test_joint_five_restart_acquisition_problem_same_path_as_scipy(native)
        test_abnormal_line_search_and_memoised_repeats_like_scipy(native)
        # and it IS another order: on a long sum-dominated problem the two orders part in the last bits
        rng = np.random.default_rng(3)
        n = 150
        x0, lo, hi = rng.uniform(-2, 2, n), np.full(n, -3.0), np.full(n, 3.0)
        tree = native.lbfgsb_minimize(rosen, x0, list(zip(lo, hi)), maxiter=60)
        native.lbfgsb_set_sum_order(0)
        seq = native.lbfgsb_minimize(rosen, x0, list(zip(lo, hi)), maxiter=60)
        assert not np.array_equal(tree["x"], seq["x"])
        assert np.abs(tree["x"] - seq["x"]).max() < 1e-3          # (60 unconverged iterations amplify the last bit: tests/test_lbfgsb_divergence.py)
    finally:
        native.lbfgsb_set_sum_order(was)
