"""Sweeps / rounds / time of the Jacobi kernel inside the real headline run (timing build)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PCABO_LIB"] = os.path.join(ROOT, "para-ortho-pca-bo_amd", "lib", "libpcabo_timing.so")
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
from pcabo import _native as N
from Algorithms import PCA_BO
from pcabo.bbob import BBOBProblem
opt = PCA_BO(budget=450, n_DoE=120, random_seed=15400, maximization=False)
prob = BBOBProblem(15, 0, 40)
opt._start(prob)
hist = {}
tot = 0.0
NIT = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for it in range(NIT):
    opt._bo_iteration(prob)
    st = (C.c_ulonglong * 8)(); assert N.LIB.pcabo_debug_jacobi_stamps(st) == 0
    cyc, wall, sweeps, rounds = [int(v) for v in st[:4]]
    ph = [int(v) / (sweeps * rounds) for v in st[4:]]
    hist[sweeps] = hist.get(sweeps, 0) + 1
    tot += wall * 0.01
    if it % 20 == 0:
        print(f"it {it}: sweeps {sweeps}, {wall*0.01:.1f} us, {cyc/(sweeps*rounds):.0f} cycles/round; per round: loads+dots {ph[0]:.0f}, rotation parameters {ph[1]:.0f}, apply+write {ph[2]:.0f}, barrier {ph[3]:.0f}", flush=True)
opt._finish()
print("sweeps histogram", sorted(hist.items()), "mean us", tot / NIT)
