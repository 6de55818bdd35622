"""Where does one BO iteration of the headline run spend its host time?  (diagnostic: wraps the calls of the iteration
with perf_counter; the rest is Python glue)."""
import os, sys, time
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np
import torch
from Algorithms import PCA_BO
import importlib
mod = importlib.import_module("Algorithms.BayesianOptimization.PCA_BO")
from pcabo import _native, initializers as _init, acqopt as _acqopt
from pcabo.bbob import BBOBProblem

acc = defaultdict(float)


def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t
    setattr(obj, name, w)


opt = PCA_BO(budget=450, n_DoE=120, random_seed=15400, maximization=False)
prob = BBOBProblem(15, 0, 40)
opt._start(prob)
for _ in range(5):
    opt._bo_iteration(prob)
ctx = opt._PCA_BO__ctx
for nm in ("wpca_gp_condition", "wpca", "gp_condition", "acq_bounds", "gp_wait", "inverse_map", "optimize_acqf", "acq_eval"):
    wrap(ctx, nm, "ctx." + nm)
wrap(_init, "scrambled_sobol_engine"); wrap(_init, "draw_sobol")
for nm in ("_design_matrix", "_calculate_ranks", "_take_noise", "_prefetch_noise", "_transform_points_to_reduced_space",
           "_initialize_model", "optimize_acqf_and_get_observation", "_transform_point_to_original_space", "assign_new_best"):
    if hasattr(opt, nm):
        wrap(opt, nm)
wrap(mod._acqopt, "optimize_acqf", "acqopt.optimize_acqf")
N = 320
t0 = time.perf_counter()
for _ in range(N):
    opt._bo_iteration(prob)
tot = time.perf_counter() - t0
opt._finish()
print(f"iteration {tot / N * 1e3:.3f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:42s} {v / N * 1e3:8.3f} ms")
