"""Which part of a preceding (warm-up) run makes the initial pick of the next run slow?  (diagnostic)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np, torch
from Algorithms import PCA_BO
from pcabo import _native
from pcabo.bbob import BBOBProblem
mode = sys.argv[1]
torch.set_num_threads(4)


def mk(inst):
    o = PCA_BO(budget=450, n_DoE=120, random_seed=15400 + inst, maximization=False)
    p = BBOBProblem(15, inst, 40)
    o._start(p)
    return o, p


keep = None
if mode == "run_close":
    o, p = mk(29)
    for _ in range(3): o._bo_iteration(p)
    o._finish()
elif mode == "run_keep":
    o, p = mk(29)
    for _ in range(3): o._bo_iteration(p)
    keep = (o, p)
elif mode == "start_close":
    o, p = mk(29)
    o._finish()
elif mode == "ctx_only":
    c = _native.Context(max_n=450, max_d=40, max_q=512); c.close()
elif mode == "run_close_del":
    o, p = mk(29)
    for _ in range(3): o._bo_iteration(p)
    o._finish(); del o, p
    import gc; gc.collect()
o2, p2 = mk(0)
for _ in range(5): o2._bo_iteration(p2)
o2.phase_breakdown.update({k: 0.0 for k in o2.phase_breakdown})
t0 = time.perf_counter()
N = 150
for _ in range(N): o2._bo_iteration(p2)
tot = time.perf_counter() - t0
pb = o2.phase_breakdown
print(f"{mode:14s} iteration {tot/N*1e3:.3f} ms  pick {pb['init_pick']/N*1e3:.3f}  raw {pb['raw_eval']/N*1e3:.3f}")
o2._finish()
