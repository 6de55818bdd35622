import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "para-ortho-pca-bo_amd"), os.path.join(ROOT, "oracle")): sys.path.insert(0, p)
import pcabo_oracle as O
from pcabo.bbob import BBOBProblem
prob = BBOBProblem(15, 0, 40)
base = O.OraclePCABO(budget=450, n_DoE=120, random_seed=15400)
base.initial_design(prob, 40, np.full(40, -5.0), np.full(40, 5.0))
rng = np.random.default_rng(0)
while len(base.x_evals) < 300:                       # synthetic state at n=300
    x = rng.uniform(-5, 5, 40); base.x_evals.append(x); base.f_evals.append(prob(x))
base._assign_new_best()
for nt in (1, 2, 4, 8, 16, 32):
    torch.set_num_threads(nt)
    ts = []
    for rep in range(2):
        o = O.OraclePCABO(budget=450, n_DoE=120, random_seed=15400)
        o.x_evals = [v.copy() for v in base.x_evals]; o.f_evals = list(base.f_evals); o._assign_new_best()
        np.random.seed(1); torch.manual_seed(1)
        t = time.perf_counter(); o.step(prob, np.full(40, -5.0), np.full(40, 5.0)); ts.append(time.perf_counter() - t)
    print(f"threads={nt:2d}: {min(ts):.3f} s per oracle BO iteration at n=300", flush=True)
