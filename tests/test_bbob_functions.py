"""BBOB f16-f19, f21-f24 (BASELINE.json configs[2] / [3]) - restated from the published COCO legacy definitions and
**parity unpinned**: the reference commits runs of f15 and f20 only (those two are pinned in test_reference_kats.py).
What can be checked without known answers: the defining properties of each function."""
import math

import numpy as np
import pytest

from pcabo import bbob
from pcabo.bbob import BBOBProblem, FUNCTIONS

UNPINNED = [f for f in sorted(FUNCTIONS) if f not in bbob.PINNED_BY_REFERENCE_DATA]


def test_function_table_covers_configs_2_and_3():
    assert sorted(FUNCTIONS) == list(range(15, 25))
    assert bbob.PINNED_BY_REFERENCE_DATA == (15, 20) and UNPINNED == [16, 17, 18, 19, 21, 22, 23, 24]
    names = {fid: FUNCTIONS[fid][0] for fid in FUNCTIONS}
    assert names[16] == "Weierstrass" and names[21] == "Gallagher101" and names[24] == "LunacekBiRastrigin"


@pytest.mark.parametrize("fid", sorted(FUNCTIONS))
@pytest.mark.parametrize("dim", [2, 5, 20, 40])
def test_optimum_is_zero_at_xopt_and_positive_elsewhere(fid, dim):
    rng = np.random.default_rng(100 * fid + dim)
    for inst in (0, 7):
        p = BBOBProblem(fid, inst, dim)
        xo = p.optimum.x
        assert xo.shape == (dim,) and np.all(np.abs(xo) <= 5.0)
        assert abs(p.raw(xo)) < 1e-9                                  # raw value = f - f_opt
        assert p(xo) == pytest.approx(p.f_opt, abs=1e-9)
        vals = np.array([p.raw(rng.uniform(-5, 5, dim)) for _ in range(50)])
        assert np.all(vals > 0.0) and np.all(np.isfinite(vals))
        near = np.array([p.raw(xo + 1e-3 * rng.normal(size=dim)) for _ in range(10)])
        assert np.all(near >= 0.0) and near.max() < np.median(vals)   # a basin around x_opt
        assert isinstance(p(rng.uniform(-5, 5, dim)), float)          # Python float, like ioh (best_f stays float32-rounded)


@pytest.mark.parametrize("fid", UNPINNED)
def test_deterministic_per_instance_and_distinct_between_instances(fid):
    x = np.linspace(-4, 4, 10)
    a, b, c = BBOBProblem(fid, 3, 10), BBOBProblem(fid, 3, 10), BBOBProblem(fid, 4, 10)
    assert a.raw(x) == b.raw(x) and a.raw(x) != c.raw(x)
    assert not np.array_equal(a.optimum.x, c.optimum.x)
    assert -1000.0 <= a.f_opt <= 1000.0 and round(a.f_opt * 100) == pytest.approx(a.f_opt * 100, abs=1e-9)


def test_boundary_penalty_outside_the_box():
    x = np.zeros(6)
    x[2] = 6.5                                                   # 1.5 outside: f_pen = 2.25
    for fid, factor in ((16, 10.0 / 6), (17, 10.0), (18, 10.0), (21, 1.0), (22, 1.0), (23, 1.0), (24, 1e4)):
        p = BBOBProblem(fid, 1, 6)
        xi = x.copy()
        xi[2] = 5.0
        # value outside >= penalty alone; and the penalty term is what distinguishes the two evaluations' lower bound
        assert p.raw(x) >= factor * 2.25 - 1e-9, fid


def test_f18_is_f17_with_conditioning_1000_on_the_same_seed():
    a, b = BBOBProblem(17, 2, 8), BBOBProblem(18, 2, 8)
    assert np.array_equal(a.optimum.x, b.optimum.x)              # COCO: rseed_17 for both
    assert np.array_equal(a._state.rot_r, b._state.rot_r)
    assert a.f_opt == b.f_opt
    x = np.full(8, 1.5)
    assert a.raw(x) != b.raw(x)
    assert b._state.scale[-1] == pytest.approx(math.sqrt(1000.0)) and a._state.scale[-1] == pytest.approx(math.sqrt(10.0))


def test_gallagher_structure():
    for fid, peaks, first in ((21, 101, math.sqrt(1000.0)), (22, 21, 1000.0)):
        st = BBOBProblem(fid, 0, 10)._state
        assert st.centres.shape == (peaks, 10) and st.heights[0] == 10.0
        assert st.heights[1] == pytest.approx(1.1) and st.heights[-1] == pytest.approx(9.1)
        # per peak: a permutation of cond^(j/(D-1) - 1/2)
        assert np.sort(st.scales[0]) == pytest.approx(first ** (np.arange(10) / 9.0 - 0.5))
        conds = np.sort(st.scales[1:].max(axis=1) ** 2)          # max scale = cond^(1/2)
        assert conds == pytest.approx(1000.0 ** (np.arange(peaks - 1) / (peaks - 2.0)))
        # every local optimum is a stationary bump: the value at a centre is governed by that peak's height or a higher one
        p = BBOBProblem(fid, 0, 10)
        x1 = st.rot.T @ st.centres[peaks - 1]
        assert p.raw(x1) <= (10.0 - 9.1) ** 2 * 1.5 + bbob._boundary_penalty(x1)


def test_rotations_are_orthogonal_and_m_has_the_stated_conditioning():
    for fid, cond in ((16, 100.0), (23, 100.0)):
        st = BBOBProblem(fid, 5, 12)._state
        assert np.abs(st.rot_r @ st.rot_r.T - np.eye(12)).max() < 1e-12
        sv = np.linalg.svd(st.m, compute_uv=False)
        assert (sv.max() / sv.min()) ** 2 == pytest.approx(cond, rel=1e-9)


def test_unknown_function_raises():
    with pytest.raises(NotImplementedError):
        BBOBProblem(14, 0, 5)
