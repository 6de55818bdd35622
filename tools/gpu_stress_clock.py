"""BASELINE.json configs[4] end to end: PCA_BO on BBOB f15, d=100, doe_factor=3 (n_DoE 300), budget_factor=10 (budget 1050),
256 EI multi-starts (torch_config["NUM_RESTARTS"] = 256; 52 joint L-BFGS-B problems of 5).  Times every BO iteration of the
run (or the first `max_seconds`), prints one JSON line.  usage: gpu_stress_clock.py [max_seconds] [num_restarts] [profile]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np
import torch
from Algorithms import PCA_BO
from pcabo.bbob import BBOBProblem
max_s = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
nres = int(sys.argv[2]) if len(sys.argv) > 2 else 256
prof = len(sys.argv) > 3 and sys.argv[3] == "profile"
torch.set_num_threads(4)
prob = BBOBProblem(15, 0, 100)
opt = PCA_BO(budget=1050, n_DoE=300, random_seed=16000, maximization=False)
opt.torch_config["NUM_RESTARTS"] = nres
opt._start(prob)
if prof:
    opt.device_context.set_profiling(True)
t_it, ns, ks = [], [], []
t0 = time.perf_counter()
while opt.number_of_function_evaluations < opt.budget and time.perf_counter() - t0 < max_s:
    ns.append(len(opt.f_evals))
    a = time.perf_counter()
    opt._bo_iteration(prob)
    t_it.append(time.perf_counter() - a)
    ks.append(int(opt.reduced_space_dim_num))
dev = opt.device_context.profile() if prof else None
phases = dict(opt.total_times); phases.update({"optimize_acqf/" + k: v for k, v in opt.phase_breakdown.items()})
rounds = int(sum(int(i[:, 1].max()) for i in opt.lbfgsb_info))
opt._finish()
t_it, ns = np.array(t_it), np.array(ns)
out = {"config": "configs[4]: f15 d=100, n_DoE 300, budget 1050, %d multi-starts" % nres, "iterations_timed": len(t_it),
       "n_range": [int(ns[0]), int(ns[-1])], "k_range": [min(ks), max(ks)], "seconds": float(t_it.sum()),
       "bo_iterations_per_s": len(t_it) / float(t_it.sum()), "ms_per_iteration_mean": 1e3 * float(t_it.mean()),
       "ms_by_n": {f"{lo}-{hi}": 1e3 * float(t_it[(ns >= lo) & (ns < hi)].mean()) for lo, hi in ((300, 450), (450, 600), (600, 800), (800, 1051))
                   if ((ns >= lo) & (ns < hi)).any()},
       "host_phase_seconds": phases, "lbfgsb_rounds": rounds, "best_f": float(min(opt.f_evals)),
       "device_profile": dev}
print(json.dumps(out))
