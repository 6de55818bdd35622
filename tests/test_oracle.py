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


def _logged_candidate_cases(G, every=6):
    for run in G["vanilla_runs"]:
        rows = np.array(run["rows"])
        for t in range(10, 75, every):
            yield run["fid"], run["instance"], t, rows[:t, 1:], rows[:t, 0], rows[t, 1:]


def test_reference_logged_vanilla_candidates_are_optima_of_the_oracle_surface():
    """Pin of rows F-L by the reference's OWN outputs: every BO row of the committed Vanilla_BO runs is the point the
    reference's optimize_acqf returned for the GP built from the rows before it, so it has to be a local maximum of
    the oracle's log-EI surface (x is printed to 1e-6, L-BFGS-B stops at a relative f-reduction of 2e-9).
    The check has power: with lengthscale 1.0 instead of ln 2, or noise 1e-4 instead of e^-5, the same points are
    NOT optima (moves of 0.5 / 3e-3 instead of 2e-5)."""
    import json, os
    from scipy.optimize import minimize
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kats_dim5.json")))
    assert len(G["vanilla_runs"]) == 12
    ident = np.vstack([np.zeros(5), np.ones(5)])

    def moves(lengthscale, noise, every):
        dx, dv = [], []
        for fid, inst, t, X, f, xc in _logged_candidate_cases(G, every):
            gp = O.ExactGP(X, f, ident, lengthscale=lengthscale, noise=noise)
            acq = O.Acquisition(gp, float(f.min()), False, "expected_improvement")

            def fg(x):
                v, g = acq.value_and_grad(x.reshape(1, -1))
                return -float(v[0]), -np.asarray(g).ravel()
            v0 = -fg(xc)[0]
            res = minimize(fg, xc, jac=True, method="L-BFGS-B", bounds=[(-5, 5)] * 5,
                           options=dict(maxiter=200, ftol=1e7 * np.finfo(float).eps, gtol=1e-5))
            dx.append(np.abs(res.x - xc).max())
            dv.append(-res.fun - v0)
        return np.array(dx), np.array(dv)

    dx, dv = moves(O.LENGTHSCALE, O.NOISE, 6)
    assert len(dx) == 12 * 11
    assert np.median(dx) < 1e-4 and np.quantile(dx, 0.9) < 5e-4 and dx.max() < 5e-3, (np.median(dx), dx.max())
    assert np.median(dv) < 1e-8 and np.quantile(dv, 0.9) < 1e-7
    dx_ls, _ = moves(1.0, O.NOISE, 13)
    dx_noise, _ = moves(O.LENGTHSCALE, 1e-4, 13)
    assert np.median(dx_ls) > 0.1 and np.median(dx_noise) > 10 * np.median(dx)


def test_all_sixty_reference_vanilla_runs_and_the_remembered_constants():
    """The local-optimum pin over ALL 60 committed Vanilla_BO runs (three BO rows of each here; every row on the device in
    tests/test_gpu_parity.py), and one sensitivity row per constant that comes from memory of the absent packages: does the
    reference's own data pin it (the logged candidates stop being optima when it is changed), or is it listed as
    unpinned together with its measured effect?"""
    import json, os
    from scipy.optimize import minimize
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_vanilla_runs_dim5.json")))
    assert len(G["vanilla_runs"]) == 60 and all(len(r["rows"]) == 75 for r in G["vanilla_runs"])
    ident = np.vstack([np.zeros(5), np.ones(5)])

    def cases(rows_at):
        for run in G["vanilla_runs"]:
            rows = np.array(run["rows"])
            for t in rows_at:
                yield rows[:t, 1:], rows[:t, 0], rows[t, 1:]

    def moves(rows_at, **gp_kw):
        dx = []
        for X, f, xc in cases(rows_at):
            gp = O.ExactGP(X, f, ident, **gp_kw)
            acq = O.Acquisition(gp, float(f.min()), False, "expected_improvement")

            def fg(x):
                v, g = acq.value_and_grad(x.reshape(1, -1))
                return -float(v[0]), -np.asarray(g).ravel()
            res = minimize(fg, xc, jac=True, method="L-BFGS-B", bounds=[(-5, 5)] * 5,
                           options=dict(maxiter=200, ftol=1e7 * np.finfo(float).eps, gtol=1e-5))
            dx.append(np.abs(res.x - xc).max())
        return np.array(dx)

    dx = moves((17, 41, 68))
    assert len(dx) == 180
    assert np.median(dx) < 1e-4 and np.quantile(dx, 0.9) < 1e-3 and np.mean(dx < 5e-3) >= 0.97, (np.median(dx), np.sort(dx)[-6:])
    base = moves((41,))
    # BEST_F_FLOAT32: float32 rounding of best_f moves an optimum by less than the 1e-6 the files print -> NOT pinned by
    # the reference's data (effect measured here: the two variants' optima agree to < 1e-5)
    try:
        O.BEST_F_FLOAT32 = False
        assert np.abs(moves((41,)) - base).max() < 1e-4
    finally:
        O.BEST_F_FLOAT32 = True
    # LOG_EI_U_EPS_CLAMP: only reached for u < -1e6, never on these surfaces -> not pinned; identical optima
    try:
        O.LOG_EI_U_EPS_CLAMP = False
        assert np.array_equal(moves((41,))[:20], base[:20])
    finally:
        O.LOG_EI_U_EPS_CLAMP = True
    # INIT_ETA (temperature of the Boltzmann pick) does not enter a local-optimum check at all: it selects WHICH optimum a
    # restart starts near.  Not pinned; its effect is the choice among restarts, covered only in distribution by the
    # end-to-end check against the reference's final results (tests/test_gpu_parity.py).
    assert O.INIT_ETA == 1.0
    # pinned constants, for contrast: the GP noise and the lengthscale DO move the optima far beyond print precision
    assert np.median(moves((41,), noise=1e-4)[:20]) > 10 * np.median(base[:20])
    assert np.median(moves((41,), lengthscale=1.0)[:20]) > 0.1
