"""Soak run (diagnostic): full-budget PCA_BO runs over several dimensions / instances; reports it/s, optimiser
warnflags, retries and out-of-bounds counts."""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
from Algorithms import PCA_BO
from pcabo.bbob import BBOBProblem

cases = [(10, i, None) for i in range(3)] + [(20, i, None) for i in range(3)] + [(40, i, None) for i in (1, 2)] + [(100, 0, 60)]
for dim, inst, cap in cases:
    budget, ndoe = 10 * dim + 50, 3 * dim
    if cap:
        budget = ndoe + cap
    prob = BBOBProblem(15, inst, dim)
    opt = PCA_BO(budget=budget, n_DoE=ndoe, random_seed=15000 + 10 * dim + inst, maximization=False)
    with warnings.catch_warnings(record=True) as ws:
        warnings.simplefilter("always")
        t = time.perf_counter()
        opt(prob)
        dt = time.perf_counter() - t
    iters = budget - ndoe
    flags = np.concatenate([i[:, 2] for i in opt.lbfgsb_info])
    tasks = np.concatenate([i[:, 3] for i in opt.lbfgsb_info])
    f = np.array(opt.f_evals)
    print(f"d={dim:3d} inst={inst} iters={iters:3d} {iters / dt:7.1f} it/s  best={opt.current_best:.4f} "
          f"oob={(f[ndoe:] == 1000).sum():3d} warnflag1={(flags == 1).sum()} warnflag2={(flags == 2).sum()} "
          f"abnormal={(tasks == 70).sum()} retries={sum('trying again' in str(w.message) for w in ws)} "
          f"k_last={opt.reduced_space_dim_num} phase={ {k: round(v, 2) for k, v in opt.total_times.items()} }", flush=True)
