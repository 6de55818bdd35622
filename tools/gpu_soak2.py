import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
from Algorithms import PCA_BO
from pcabo.bbob import BBOBProblem
for dim, inst in ((40, 0), (40, 1)):
    budget, ndoe = 10 * dim + 50, 3 * dim
    prob = BBOBProblem(15, inst, dim)
    opt = PCA_BO(budget=budget, n_DoE=ndoe, random_seed=15000 + 10 * dim + inst, maximization=False, record_trace=True)
    t = time.perf_counter(); opt(prob); dt = time.perf_counter() - t
    ks = np.array([tr["k"] for tr in opt.trace])
    rounds = np.array([int(i[:, 1].max()) for i in opt.lbfgsb_info])
    nits = np.array([i[:, 0].tolist() for i in opt.lbfgsb_info])
    tt = np.array(opt.timing_logs["optimize_acqf"])
    print(f"d={dim} inst={inst}: {len(ks)/dt:.1f} it/s; k: first {ks[:5].tolist()} .. at 50/100/200/329: {ks[[50,100,200,329]].tolist()}")
    for a, b in ((0, 50), (50, 100), (100, 200), (200, 330)):
        print(f"   iters {a:3d}-{b:3d}: mean k {ks[a:b].mean():5.1f}  mean rounds/iter {rounds[a:b].mean():7.1f}  mean L-BFGS-B its {nits[a:b].mean():6.1f} "
              f" optimize ms/iter {1e3*tt[a:b].mean():6.2f}  us/round {1e6*tt[a:b].sum()/rounds[a:b].sum():6.1f}  breakdown {opt.phase_breakdown}")
