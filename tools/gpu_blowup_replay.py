"""A run whose search box blows up (the reference keeps out-of-box candidates: PCA_BO.py:253,260-263): f21 / instance 25 /
d = 40 by default.  Runs it on the device, then teacher-forces the oracle from the device's states over the last
iterations (and at the first state with |x| > 1e70) and prints what both produce.  Round 3 diagnostic."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("para-ortho-pca-bo_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import pcabo_oracle as O
from pcabo import _native as N
from pcabo.bbob import BBOBProblem
from Algorithms import PCA_BO

torch.set_num_threads(4)
fid, inst, dim = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (21, 25, 40)))
seed = 1000 * fid + 10 * dim + inst
opt = PCA_BO(budget=10 * dim + 50, n_DoE=3 * dim, random_seed=seed, maximization=False, record_trace=True, acq_kernel="group")
err = None
try:
    opt(BBOBProblem(fid, inst, dim))
except N.PcaboError as e:
    err = e
n_end = len(opt.f_evals)
X, f = np.vstack(opt.x_evals), np.array(opt.f_evals, dtype=float)
print(f"device run: {'completed' if err is None else 'STOPPED: ' + str(err)} with {n_end} evaluations; max |x| = {np.abs(X).max():.3e}; "
      f"penalised {int((f == 1000.0).sum())}; best {f.min():.6g}")
big = [t["n"] for t in opt.trace if np.abs(X[:t['n']]).max() > 1e70]
pick = sorted(set(([big[0]] if big else []) + [t["n"] for t in opt.trace[-6:]]))
for tr in opt.trace:
    if tr["n"] not in pick:
        continue
    n = tr["n"]
    orc = O.OraclePCABO(budget=n + 1, n_DoE=n, random_seed=0, maximization=False, record=True)
    orc.x_evals = [r.copy() for r in X[:n]]; orc.f_evals = [float(v) for v in f[:n]]; orc._assign_new_best()
    np.random.set_state(tr["numpy_state"]); torch.set_rng_state(tr["torch_state"])
    try:
        rec = orc.step(BBOBProblem(fid, inst, dim), np.full(dim, -5.0), np.full(dim, 5.0))
    except Exception as e:   # noqa: BLE001
        print(f"n={n}: ORACLE RAISED {e!r}; device: {'has a candidate' if n < n_end else 'stopped here too'}")
        continue
    line = f"n={n}: max|x| {np.abs(X[:n]).max():.2e} oracle k {rec.k} device k {tr.get('k')}"
    if "cands" in tr:
        same_picks = sorted(rec.trace.ic_idx.tolist()) == sorted(tr["ic_idx"].tolist())
        sc = max(1.0, np.abs(rec.trace.cands).max())
        dc = np.abs(rec.trace.cands - tr["cands"]).max(axis=1) / sc
        dx = np.abs(rec.cand_x - X[n]).max() / max(1.0, np.abs(rec.cand_x).max()) if n < n_end else float("nan")
        vo = rec.acq(torch.from_numpy(np.ascontiguousarray(tr["cands"]))).detach().numpy()
        ds = float((np.abs(vo - tr["vals"]) / np.maximum(1.0, np.abs(tr["vals"]))).max())
        line += f" same picks {same_picks}; end points max rel diff {dc.max():.2e} median {np.median(dc):.2e}; chosen x rel diff {dx:.2e}; device surface judged by oracle {ds:.2e}; oracle L-BFGS-B {[(t.nit, t.nfev) for t in rec.trace.lbfgsb]} device {tr['info'][:, :2].tolist()}"
    else:
        line += "  (the device stopped in this iteration)"
    print(line, flush=True)
