"""Diagnostic: where a GPU run and its oracle replay disagree on botorch's retry, show how close the case is."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "para-ortho-pca-bo_amd"), os.path.join(ROOT, "oracle")): sys.path.insert(0, p)
import pcabo_oracle as O
from pcabo.bbob import BBOBProblem
from pcabo import _native as N
from Algorithms import PCA_BO
torch.set_num_threads(4)
opt = PCA_BO(budget=150, n_DoE=30, random_seed=15100, maximization=False, record_trace=True)
opt(BBOBProblem(15, 0, 10))
X_all, f_all = np.vstack(opt.x_evals), np.array(opt.f_evals, dtype=float)
for it, tr in enumerate(opt.trace):
    n = tr["n"]
    orc = O.OraclePCABO(budget=n + 1, n_DoE=n, random_seed=0, record=True)
    orc.x_evals = [r.copy() for r in X_all[:n]]; orc.f_evals = [float(v) for v in f_all[:n]]; orc._assign_new_best()
    np.random.set_state(tr["numpy_state"]); torch.set_rng_state(tr["torch_state"])
    rec = orc.step(BBOBProblem(15, 0, 10), np.full(10, -5.0), np.full(10, 5.0))
    g_ret = bool(tr.get("retried", False))
    if rec.trace.retried != g_ret:
        print(f"it={it} n={n} k={rec.k} oracle retried={rec.trace.retried} gpu retried={g_ret}")
        print("  oracle lbfgsb:", [(t.nit, t.nfev, t.status, t.message[:30]) for t in rec.trace.lbfgsb])
        print("  gpu info:", tr["info"].tolist())
        # evaluate both gradients at the oracle's first initial conditions
        gp = O.ExactGP(rec.wpca.Z, rec.f, rec.norm_bounds); acq = O.Acquisition(gp, rec.best_f, False)
        ics = rec.trace.ics
        v, g = acq.value_and_grad(ics)
        c = N.Context(max_n=160, max_d=10, max_q=16)
        c.gp_condition(rec.f, Z=rec.wpca.Z)
        gv, gg = c.acq_eval(ics, O.round_best_f(rec.best_f), False, N.ACQ_LOG_EI, grad=True)
        print("  oracle value", v[:5], "\n  gpu value   ", gv[:5])
        print("  oracle grad", np.asarray(g).ravel()[:10], "\n  gpu grad   ", gg.ravel()[:10])
        print("  acq bounds", rec.acq_bounds.ravel(), " ics", ics.ravel())
        # the same joint 5-restart problem three ways: scipy on the oracle surface (= the replay), this library's
        # L-BFGS-B on the oracle surface, this library's L-BFGS-B on the device surface
        from scipy.optimize import minimize
        x0 = ics[:5].ravel().copy(); kk = ics.shape[1]
        bnds = [(rec.acq_bounds[0, i % kk], rec.acq_bounds[1, i % kk]) for i in range(x0.size)]
        bf = O.round_best_f(rec.best_f)
        log = {"o": [], "d": []}
        def f_or(x):
            v, g = acq.value_and_grad(x.reshape(-1, kk)); log["o"].append(-float(v.sum())); return -float(v.sum()), -np.asarray(g).ravel()
        def f_dev(x):
            v, g = c.acq_eval(x.reshape(-1, kk), bf, False, N.ACQ_LOG_EI, grad=True); log["d"].append(-float(v.sum())); return -float(v.sum()), -g.ravel()
        r = minimize(f_or, x0, jac=True, method="L-BFGS-B", bounds=bnds, options=dict(maxiter=200))
        print("  scipy/oracle  :", r.nit, r.nfev, r.status, r.message, [repr(v) for v in log["o"]]); log["o"].clear()
        r2 = N.lbfgsb_minimize(f_or, x0, bnds, maxiter=200)
        print("  native/oracle :", r2["nit"], r2["nfev"], r2["task"], [repr(v) for v in log["o"]])
        r3 = N.lbfgsb_minimize(f_dev, x0, bnds, maxiter=200)
        print("  native/device :", r3["nit"], r3["nfev"], r3["task"], [repr(v) for v in log["d"]])
        for label, sel in (("all 10", slice(0, 10)), ("first 5", slice(0, 5)), ("last 5", slice(5, 10))):
            cand, vals, info, failed = c.optimize_acqf(ics[sel], rec.acq_bounds, bf)
            print(f"  ctx.optimize_acqf {label}: failed={failed} info={info.tolist()} vals={vals.tolist()}")
        c.close()
print("done")
