"""Per-instance speed of the headline configuration (diagnostic): how uneven is 'one run per rank'?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np, torch
from Algorithms import PCA_BO
from pcabo.bbob import BBOBProblem
torch.set_num_threads(4)
for inst in range(8):
    opt = PCA_BO(budget=450, n_DoE=120, random_seed=15400 + inst, maximization=False)
    p = BBOBProblem(15, inst, 40)
    opt._start(p)
    t = time.perf_counter()
    for _ in range(330): opt._bo_iteration(p)
    dt = time.perf_counter() - t
    rounds = sum(int(i[:, 1].max()) for i in opt.lbfgsb_info)
    opt._finish()
    print(f"instance {inst}: {330/dt:6.1f} it/s, {rounds} L-BFGS-B rounds, best {opt.current_best:.3f}", flush=True)
