"""Can the reference's committed PCA_BO logs (pca-experiment/, an OLDER revision that clipped candidates - SURVEY.md fact 6)
pin the current PCA chain after all?  Probe, run in the build container only (reads /root/reference):
every BO row that lies strictly inside the box (535 of 3 900) would have to lie in the affine subspace of the weighted
PCA of the rows before it.  Result: it does not - residuals of 0.1-0.5 for k = 3, 4 under rank weights (either direction),
uniform weights and all three mean conventions; only k = 5 (no reduction) fits trivially.  The older revision's PCA step
differs from the current one, so those rows pin nothing of rows A-C."""
import sys, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import pcabo_oracle as O
from scipy.optimize import minimize
REF = "/root/reference"
def read_runs(path):
    runs, cur = [], None
    for line in open(path):
        t = line.split()
        if not t: continue
        if t[0] == "evaluations":
            cur = []; runs.append(cur)
        else: cur.append([float(v) for v in t])
    return runs
stats = []
for fid, name in ((15, "RastriginRotated"), (20, "Schwefel")):
    runs = read_runs(f"{REF}/pca-experiment/data_f{fid}_{name}/IOHprofiler_f{fid}_DIM5.dat")
    for inst, run in enumerate(runs):
        rows = np.array(run)    # evaluations raw_y raw_y_best x0..x4
        X, f = rows[:, 3:], rows[:, 1]
        for t in range(10, len(rows)):
            xc = X[t]
            interior = np.all(np.abs(xc) < 5 - 1e-6)
            if not interior: continue
            wp = O.weighted_pca(X[:t], list(f[:t]), False, 0.95, 0, noise=np.zeros((t, 5)))
            ck = wp.components[:wp.k]
            off = xc - wp.data_mean - wp.pca_mean
            z = off @ ck.T
            resid = np.abs(off - z @ ck).max()
            stats.append((fid, inst, t, wp.k, resid, z))
print("interior rows:", len(stats))
ks = np.array([s[3] for s in stats]); res = np.array([s[4] for s in stats])
for k in range(1, 6):
    m = ks == k
    if m.any(): print("k =", k, "rows", m.sum(), "subspace residual median %.2e q90 %.2e max %.2e" % (np.median(res[m]), np.quantile(res[m], .9), res[m].max()))

# variants
def resid_for(Xp, fp, xc, weights_mode, kk):
    n = len(fp)
    if weights_mode == "rank": w = O.calculate_weights(list(fp), False)
    elif weights_mode == "rank_max": w = O.calculate_weights(list(fp), True)
    else: w = np.full(n, 1.0 / n)
    mu = Xp.mean(0)
    xcn = Xp - mu
    wx = xcn * np.sqrt(w[:, None])
    comps, evr, pmean = O.pca_fit(wx, True)
    out = {}
    for k in range(1, 6):
        ck = comps[:k]
        for meanmode in ("both", "data", "wmean"):
            if meanmode == "both": off = xc - mu - pmean
            elif meanmode == "data": off = xc - mu
            else: off = xc - (w[:, None] * Xp).sum(0)
            out[(k, meanmode)] = np.abs(off - (off @ ck.T) @ ck).max()
    return out, evr
import collections
agg = collections.defaultdict(list)
for fid, name in ((15, "RastriginRotated"),):
    runs = read_runs(f"{REF}/pca-experiment/data_f{fid}_{name}/IOHprofiler_f{fid}_DIM5.dat")
    for inst, run in enumerate(runs[:30]):
        rows = np.array(run); X, f = rows[:, 3:], rows[:, 1]
        for t in range(10, len(rows)):
            xc = X[t]
            if not np.all(np.abs(xc) < 5 - 1e-6): continue
            for wm in ("rank", "rank_max", "uniform"):
                out, evr = resid_for(X[:t], f[:t], xc, wm, 0)
                for key, v in out.items(): agg[(wm,) + key].append(v)
for key in sorted(agg, key=lambda k: np.median(agg[k]))[:12]:
    print(key, "median %.2e q90 %.2e" % (np.median(agg[key]), np.quantile(agg[key], .9)), len(agg[key]))
