"""The oracle checked against real third-party code available here (sklearn, torch autograd, finite
differences) - these are the pins the restated rows can have without botorch/gpytorch."""
import numpy as np
import pytest
import torch

import pcabo_oracle as O
from pcabo.bbob import BBOBProblem


def test_numpy_pca_restatement_equals_sklearn():
    rng = np.random.default_rng(0)
    for n, d in [(30, 10), (150, 10), (120, 40), (450, 40)]:     # both sides of sklearn's solver switch
        w = rng.normal(size=(n, d)) * rng.uniform(0.2, 2, size=d)
        a, b = O.pca_fit(w, True), O.pca_fit_numpy(w)
        for x, y in zip(a, b):
            assert np.abs(x - y).max() < 1e-12


def test_weights_follow_reference_formula():
    f = [3.0, 1.0, 2.0, 1000.0, 0.5]
    w = O.calculate_weights(f, False)
    assert w.sum() == pytest.approx(1.0)
    assert np.argmax(w) == 4 and w[3] == 0.0          # worst point gets weight 0 (PCA_BO.py:336)
    assert np.argmax(O.calculate_weights(f, True)) == 3


def test_log_ei_helper_against_naive_and_asymptote():
    u = torch.linspace(-6, 4, 401, dtype=torch.float64)
    naive = torch.log(O._phi(u) + u * O._Phi(u))
    assert (O.log_ei_helper(u) - naive).abs().max() < 1e-10
    big = torch.tensor([-50.0, -1e3, -1e5, -1e7], dtype=torch.float64)
    asym = -0.5 * big ** 2 - 0.5 * np.log(2 * np.pi) - 2 * torch.log(big.abs())
    assert ((O.log_ei_helper(big) - asym) / asym).abs().max() < 1e-3


def _small_gp(seed=0, n=40, k=4):
    rng = np.random.default_rng(seed)
    Z = rng.normal(size=(n, k))
    y = rng.normal(size=n) * 50 + 300
    y[::7] = 1000.0
    return O.ExactGP(Z, y), Z, y


def test_gp_posterior_interpolates_and_variance_is_sane():
    gp, Z, y = _small_gp()
    mean, var = gp.posterior(torch.from_numpy(Z))
    resid = np.abs(mean.numpy() - y)
    assert resid.mean() < 0.5 * y.std()                          # smoother, not interpolator (noise 0.0067)
    assert (var.numpy() > 0).all() and (var.numpy() < gp.y_std.item() ** 2).all()
    far = torch.full((1, Z.shape[1]), 50.0, dtype=torch.float64)
    m_far, v_far = gp.posterior(far)
    assert m_far.item() == pytest.approx(gp.y_mean.item(), rel=1e-9)   # prior mean 0 (standardised)
    assert v_far.item() == pytest.approx(gp.y_std.item() ** 2, rel=1e-9)


def test_acquisition_gradient_matches_finite_differences():
    gp, Z, y = _small_gp(1)
    for kind in ("expected_improvement", "probability_of_improvement"):
        acq = O.Acquisition(gp, float(y.min()), False, kind)
        X = np.random.default_rng(2).normal(size=(5, Z.shape[1]))
        v, g = acq.value_and_grad(X)
        h = 1e-6
        for j in range(Z.shape[1]):
            Xp, Xm = X.copy(), X.copy()
            Xp[:, j] += h
            Xm[:, j] -= h
            fd = (acq.value_and_grad(Xp)[0] - acq.value_and_grad(Xm)[0]) / (2 * h)
            assert np.abs(fd - g[:, j]).max() < 1e-5 * max(1.0, np.abs(g[:, j]).max())


def test_oracle_loop_is_deterministic_and_respects_oob_rule():
    torch.set_num_threads(1)
    runs = []
    for _ in range(2):
        p = BBOBProblem(15, 0, 5)
        o = O.OraclePCABO(budget=16, n_DoE=10, random_seed=15050, num_restarts=2, raw_samples=32, record=True)
        o(p, 5, np.array([-5.0, 5.0]))
        runs.append(o)
    a, b = runs
    assert len(a.f_evals) == 16
    assert np.array_equal(np.vstack(a.x_evals), np.vstack(b.x_evals))
    for rec in a.records:
        assert rec.oob == (np.any(rec.cand_x < -5) or np.any(rec.cand_x > 5))
        if rec.oob:
            assert rec.f_new == 1000.0
