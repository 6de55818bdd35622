"""Known answers taken from the reference's own committed data (tests/golden/ref_kats_dim5.json,
made by tests/golden/make_reference_kats.py from /root/reference/{pca,vanilla}-experiment):
they pin the seed formula + LHS design and the BBOB f15 objective used by oracle and product."""
import json
import os

import numpy as np
import pytest

import pcabo_oracle as O
from pcabo.bbob import BBOBProblem
from pcabo.lhs import lhs_center

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kats_dim5.json")))


def test_lhs_design_matches_all_120_reference_runs():
    assert len(G["doe"]) == 120
    for run in G["doe"]:
        assert run["seed"] == 1000 * run["fid"] + 10 * run["dim"] + run["instance"]   # ExperimentRunner.py:146
        for impl in (lhs_center, O.lhs_center):
            np.random.seed(run["seed"])
            x0 = 10.0 * impl(5, 10) - 5.0
            assert np.abs(x0 - np.array(run["x"])).max() < 5e-7      # file prints 6 decimals


def test_f15_on_exact_doe_points():
    by_inst = {(r["alg"], r["instance"]): r for r in G["f15_doe"]}
    worst = 0.0
    for run in G["doe"]:
        if run["fid"] != 15:
            continue
        np.random.seed(run["seed"])
        x0 = 10.0 * lhs_center(5, 10) - 5.0
        prob = BBOBProblem(15, run["instance"], 5)
        want = by_inst[(run["alg"], run["instance"])]["raw_y"]
        for x, y in zip(x0, want):
            worst = max(worst, abs(prob.raw(x) - y) / abs(y))
    assert worst < 5e-12


def test_f15_on_full_precision_best_points():
    worst = 0.0
    for r in G["f15_best"]:
        prob = BBOBProblem(15, r["instance"], 5)
        worst = max(worst, abs(prob.raw(np.array(r["x"])) - r["y"]) / abs(r["y"]))
    assert worst < 5e-12


def test_f15_on_printed_bo_rows():
    worst = 0.0
    for r in G["f15_bo_rows"]:
        prob = BBOBProblem(15, r["instance"], 5)
        worst = max(worst, abs(prob.raw(np.array(r["x"])) - r["raw_y"]) / abs(r["raw_y"]))
    assert worst < 5e-5           # x is printed to 1e-6 in the reference's .dat files


def test_problem_interface_and_fopt():
    p = BBOBProblem(15, 3, 40)
    assert p.meta_data.n_variables == 40 and p.meta_data.optimization_type.name == "MIN"
    assert p.bounds.lb.shape == (40,) and float(p.bounds.ub[0]) == 5.0
    x = np.zeros(40)
    assert p(x) == pytest.approx(p.raw(x) + p.f_opt)
    assert p.raw(p.optimum.x) == pytest.approx(0.0, abs=1e-9)
    with pytest.raises(NotImplementedError):
        BBOBProblem(14, 0, 5)


def test_f20_on_reference_rows():
    """BBOB f20 (Schwefel), the second function of the reference's --quick configuration: exact DoE rows,
    full-precision best points, printed BO rows."""
    by_inst = {(r["alg"], r["instance"]): r for r in G["f20_doe"]}
    worst = 0.0
    for run in G["doe"]:
        if run["fid"] != 20:
            continue
        np.random.seed(run["seed"])
        x0 = 10.0 * lhs_center(5, 10) - 5.0
        prob = BBOBProblem(20, run["instance"], 5)
        for x, y in zip(x0, by_inst[(run["alg"], run["instance"])]["raw_y"]):
            worst = max(worst, abs(prob.raw(x) - y) / max(1.0, abs(y)))
    assert worst < 1e-12
    assert len(G["f20_best"]) >= 50
    for b in G["f20_best"]:
        assert abs(BBOBProblem(20, b["instance"], 5).raw(b["x"]) - b["y"]) < 1e-12 * max(1.0, abs(b["y"]))
    bad = 0
    for r in G["f20_bo_rows"]:
        v = BBOBProblem(20, r["instance"], 5).raw(r["x"])
        bad += abs(v - r["raw_y"]) > 2e-3 * max(1.0, abs(r["raw_y"]))      # x printed with 6 decimals; f20 is steep
    assert bad <= 3
    p = BBOBProblem(20, 3, 5)
    assert abs(p.raw(p.optimum.x)) < 1e-6 and p.meta_data.name == "Schwefel"       # f(x_opt) = 0 before the shift
