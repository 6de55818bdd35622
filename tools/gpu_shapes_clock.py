"""BO iterations/s for the other shapes of BASELINE.json's configurations (one process, incl. start-up of each run)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np, torch
from Algorithms import PCA_BO, Vanilla_BO
from pcabo.bbob import BBOBProblem
torch.set_num_threads(4)
for cls, dim, ndoe, budget in ((PCA_BO, 10, 30, 150), (PCA_BO, 20, 60, 250), (PCA_BO, 40, 120, 450), (Vanilla_BO, 10, 30, 150),
                               (Vanilla_BO, 20, 60, 250)):
    its, t0 = 0, time.perf_counter()
    for inst in range(3):
        opt = cls(budget=budget, n_DoE=ndoe, random_seed=15000 + 10 * dim + inst, maximization=False)
        opt(BBOBProblem(15, inst, dim))
        its += budget - ndoe
    dt = time.perf_counter() - t0
    print(f"{cls.__name__:10s} d={dim:3d} budget {budget}: {its/dt:7.1f} BO iterations/s over instances 0-2 (incl. DoE and start-up)", flush=True)
