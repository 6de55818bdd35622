"""Distribution of GPU-vs-oracle differences over a full replayed run (diagnostic)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "para-ortho-pca-bo_amd"), os.path.join(ROOT, "oracle")): sys.path.insert(0, p)
import pcabo_oracle as O
from pcabo.bbob import BBOBProblem
from Algorithms import PCA_BO, Vanilla_BO
VANILLA = '--vanilla' in sys.argv
if VANILLA: sys.argv.remove('--vanilla')
Algo, Orc = (Vanilla_BO, O.OracleVanillaBO) if VANILLA else (PCA_BO, O.OraclePCABO)
torch.set_num_threads(4)
dim, ndoe, budget, inst = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (10, 30, 150, 0))]
seed = 15000 + 10 * dim + inst
opt = Algo(budget=budget, n_DoE=ndoe, random_seed=seed, maximization=False, record_trace=True)
opt(BBOBProblem(15, inst, dim))
X_all, f_all = np.vstack(opt.x_evals), np.array(opt.f_evals, dtype=float)
dpos, dval, dnit, dx, df, ties, kdiff, icdiff = [], [], [], [], [], 0, 0, 0
dic, dZ, opt_bounds = [], [], None
dsurf = []
for it, tr in enumerate(opt.trace):
    n = tr["n"]
    orc = Orc(budget=n + 1, n_DoE=n, random_seed=0, record=True)
    orc.x_evals = [r.copy() for r in X_all[:n]]; orc.f_evals = [float(v) for v in f_all[:n]]; orc._assign_new_best()
    np.random.set_state(tr["numpy_state"]); torch.set_rng_state(tr["torch_state"])
    rec = orc.step(BBOBProblem(15, inst, dim), np.full(dim, -5.0), np.full(dim, 5.0))
    if rec.k != tr["k"]: kdiff += 1; continue
    if rec.trace.retried or sorted(rec.trace.ic_idx.tolist()) != sorted(tr["ic_idx"].tolist()):
        icdiff += 1
        print(f"  ic-mismatch it={it} retried={rec.trace.retried} statuses={[(t.status, t.message[:20]) for t in rec.trace.lbfgsb]} gpu_info={tr['info'].tolist()}")
        continue
    dic.append(np.abs(rec.trace.ics - tr["ics"]).max() / max(1.0, np.abs(rec.trace.ics).max()))
    dZ.append(np.abs(rec.acq_bounds - opt_bounds[it]).max() / max(1.0, np.abs(rec.acq_bounds).max()) if opt_bounds else 0.0)
    scale = max(1.0, np.abs(rec.trace.cands).max())
    vo = rec.acq(torch.from_numpy(np.ascontiguousarray(tr["cands"], dtype=np.float64))).detach().numpy()
    dsurf.extend((np.abs(vo - tr["vals"]) / np.maximum(1.0, np.abs(tr["vals"]))).tolist())
    dpos.extend((np.abs(rec.trace.cands - tr["cands"]).max(axis=1) / scale).tolist())
    for r_ in range(len(tr['cands'])):
        dp_ = np.abs(rec.trace.cands[r_] - tr['cands'][r_]).max() / scale
        if dp_ > 1e-4: print(f'   it={it} restart={r_} dpos={dp_:.2e} val_o={rec.trace.vals[r_]:.12g} val_g={tr["vals"][r_]:.12g} info={tr["info"].tolist()} oracle={[(t.nit,t.nfev) for t in rec.trace.lbfgsb]}')
    dval.extend((np.abs(rec.trace.vals - tr["vals"]) / np.maximum(1.0, np.abs(rec.trace.vals))).tolist())
    dnit.extend([abs(t.nit - int(tr["info"][g, 0])) / max(1, t.nit) for g, t in enumerate(rec.trace.lbfgsb)])
    co = int(np.argmax(rec.trace.vals))
    if co != tr["chosen"]:
        ties += 1
        v = rec.trace.vals
        print("  tie at it", it, "value gap", abs(v[co] - v[tr["chosen"]]))
    else:
        dx.append(np.abs(rec.cand_x - X_all[n]).max() / max(1.0, np.abs(rec.cand_x).max()))
        df.append(abs(rec.f_new - f_all[n]) / max(1.0, abs(f_all[n])))
q = lambda a: [float(f"{v:.2e}") for v in np.quantile(np.array(a), [0.5, 0.9, 0.99, 1.0])] if len(a) else None
print(("Vanilla_BO " if VANILLA else "PCA_BO ") + f"d={dim} iters={len(opt.trace)} k-mismatch={kdiff} ic-mismatch={icdiff} argmax-ties={ties}")
print(" initial-condition position rel diff    :", q(dic))
print(" oracle acquisition AT the device's end points vs device values:", q(dsurf))
print(" restart end-point diff  (q50,q90,q99,max):", q(dpos), " fraction < 1e-5:", float(np.mean(np.array(dpos) < 1e-5)))
print(" restart value rel diff  (q50,q90,q99,max):", q(dval))
print(" L-BFGS-B iteration rel diff             :", q(dnit), " exact-equal fraction", float(np.mean(np.array(dnit) == 0)))
print(" chosen x rel diff (no-tie iterations)   :", q(dx))
print(" objective rel diff (no-tie iterations)  :", q(df))
