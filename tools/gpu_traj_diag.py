"""Free-running trajectory: GPU PCA_BO vs oracle, per-iteration differences (diagnostic)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "para-ortho-pca-bo_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import pcabo_oracle as O
from pcabo.bbob import BBOBProblem
from Algorithms import PCA_BO
torch.set_num_threads(4)
d, ndoe, iters, seed, inst = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (10, 30, 12, 15101, 1))]
o = O.OraclePCABO(budget=ndoe + iters, n_DoE=ndoe, random_seed=seed, record=True)
o(BBOBProblem(15, inst, d), d, np.array([-5.0, 5.0]))
opt = PCA_BO(budget=ndoe + iters, n_DoE=ndoe, random_seed=seed, maximization=False)
opt(BBOBProblem(15, inst, d))
Xo, Xg = np.vstack(o.x_evals), np.vstack(opt.x_evals)
for i in range(ndoe, ndoe + iters):
    rec = o.records[i - ndoe]
    info = opt.lbfgsb_info[i - ndoe].tolist()
    print(i, "dx=%.2e" % np.abs(Xo[i] - Xg[i]).max(), "f_o=%.6f f_g=%.6f" % (o.f_evals[i], opt.f_evals[i]), "k", rec.k,
          "oracle", [(t.nit, t.nfev, t.status) for t in rec.trace.lbfgsb], "gpu", [r[:3] for r in info], "retried", rec.trace.retried,
          "argmax_o", int(np.argmax(rec.trace.vals)), "vals_top2", np.sort(rec.trace.vals)[-2:].tolist())
