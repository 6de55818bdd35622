"""Cycles per Jacobi round and the shader clock actually seen by a single-work-group kernel (timing build).
usage: gpu_jacobi_clock.py [d] [n at start] [iterations]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PCABO_LIB"] = os.path.join(ROOT, "para-ortho-pca-bo_amd", "lib", "libpcabo_timing.so")
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
from pcabo import _native as N
rng = np.random.default_rng(0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3 * d
c = N.Context(max_n=n0 + 64, max_d=d, max_q=16)
X = rng.uniform(-5, 5, size=(n0, d)); f = rng.normal(size=n0)
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 12):
    X = np.vstack([X, rng.uniform(-5, 5, size=(1, d))]); f = np.append(f, rng.normal())
    ranks = np.argsort(np.argsort(f)) + 1
    c.wpca(X, ranks=ranks, noise=rng.normal(0, 1e-8, X.shape), want_Z=False)
    st = (C.c_ulonglong * 8)(); assert N.LIB.pcabo_debug_jacobi_stamps(st) == 0
    cyc, wall, sweeps, rounds = [int(v) for v in st[:4]]
    ph = [int(v) for v in st[4:8]]
    tot = max(1, sweeps * rounds)
    print(f"n={X.shape[0]} d={d}: sweeps {sweeps}, {sweeps*rounds} rounds, {wall*0.01:.1f} us in the sweeps, {cyc/tot:.0f} cycles/round "
          f"(dot {ph[0]/tot:.0f}, rotation {ph[1]/tot:.0f}, apply + write {ph[2]/tot:.0f}, barrier {ph[3]/tot:.0f}), shader clock {cyc/(wall*0.01):.0f} MHz", flush=True)
c.close()
