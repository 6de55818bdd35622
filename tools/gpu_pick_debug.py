import os, sys, hashlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "para-ortho-pca-bo_amd"), os.path.join(ROOT, "oracle")): sys.path.insert(0, p)
import pcabo_oracle as O
from pcabo.bbob import BBOBProblem
from Algorithms import PCA_BO
torch.set_num_threads(4)
opt = PCA_BO(budget=150, n_DoE=30, random_seed=15100, maximization=False, record_trace=True)
opt(BBOBProblem(15, 0, 10))
X_all, f_all = np.vstack(opt.x_evals), np.array(opt.f_evals, dtype=float)
h = lambda t: hashlib.md5(t.numpy().tobytes()).hexdigest()[:8]
for it in (48, 49, 50, 51):
    tr = opt.trace[it]; n = tr["n"]
    for rep in range(2):
        orc = O.OraclePCABO(budget=n + 1, n_DoE=n, random_seed=0, record=True)
        orc.x_evals = [r.copy() for r in X_all[:n]]; orc.f_evals = [float(v) for v in f_all[:n]]; orc._assign_new_best()
        np.random.set_state(tr["numpy_state"]); torch.set_rng_state(tr["torch_state"])
        rec = orc.step(BBOBProblem(15, 0, 10), np.full(10, -5.0), np.full(10, 5.0))
        print(f"it={it} rep={rep} state={h(tr['torch_state'])} k={rec.k}/{tr['k']} oracle idx={sorted(rec.trace.ic_idx.tolist())[:5]} gpu idx={sorted(tr['ic_idx'].tolist())[:5]} "
              f"rawX equal={np.abs(rec.trace.raw_X - 0).sum() and float(np.abs(rec.trace.raw_vals - tr['raw_vals']).max()):.2e} retried={rec.trace.retried} "
              f"next_state_after_oracle={h(torch.get_rng_state())} gpu_next_state={h(opt.trace[it+1]['torch_state'])} nlb={len(rec.trace.lbfgsb)} gpu_info={tr['info'].tolist()}")
