"""Run the same configuration several times; all trajectories must be bit-identical (diagnostic)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
from Algorithms import PCA_BO
from pcabo.bbob import BBOBProblem
torch.set_num_threads(4)
dim, ndoe, budget = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (10, 30, 150))]
runs = []
for rep in range(4):
    opt = PCA_BO(budget=budget, n_DoE=ndoe, random_seed=15000 + 10 * dim, maximization=False, record_trace=True)
    opt(BBOBProblem(15, 0, dim))
    runs.append(opt)
ref = runs[0]
for r, o in enumerate(runs[1:], 1):
    X0, X1 = np.vstack(ref.x_evals), np.vstack(o.x_evals)
    same = np.array_equal(X0, X1)
    first = None
    if not same:
        first = int(np.argmax(np.any(X0 != X1, axis=1)))
        it = first - ndoe
        t0, t1 = ref.trace[it], o.trace[it]
        print(f"run {r}: first differing evaluation {first} (iteration {it}); k {t0['k']}/{t1['k']}; raw_vals equal {np.array_equal(t0['raw_vals'], t1['raw_vals'])} "
              f"max raw diff {np.abs(t0['raw_vals'] - t1['raw_vals']).max():.3e}; idx equal {np.array_equal(t0['ic_idx'], t1['ic_idx'])}; cands maxdiff {np.abs(t0['cands'] - t1['cands']).max():.3e}; "
              f"info {t0['info'].tolist()} vs {t1['info'].tolist()}")
    print(f"run {r} identical to run 0: {same}")
import hashlib
print("trajectory md5", hashlib.md5(np.vstack(ref.x_evals).tobytes()).hexdigest(), "trace[50] idx" if len(ref.trace) > 50 else "", sorted(ref.trace[50]["ic_idx"].tolist())[:4] if len(ref.trace) > 50 else "")
