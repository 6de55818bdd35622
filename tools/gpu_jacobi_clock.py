"""Cycles per Jacobi round and the shader clock actually seen by a single-work-group kernel (timing build)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PCABO_LIB"] = os.path.join(ROOT, "para-ortho-pca-bo_amd", "lib", "libpcabo_timing.so")
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
from pcabo import _native as N
rng = np.random.default_rng(0)
d = 40
c = N.Context(max_n=450, max_d=d, max_q=16)
X = rng.uniform(-5, 5, size=(120, d)); f = rng.normal(size=120)
for it in range(12):
    X = np.vstack([X, rng.uniform(-5, 5, size=(1, d))]); f = np.append(f, rng.normal())
    ranks = np.argsort(np.argsort(f)) + 1
    c.wpca(X, ranks=ranks, noise=rng.normal(0, 1e-8, X.shape), want_Z=False)
    st = (C.c_ulonglong * 8)(); assert N.LIB.pcabo_debug_jacobi_stamps(st) == 0
    cyc, wall, sweeps, rounds = [int(v) for v in st[:4]]
    print(f"n={X.shape[0]}: sweeps {sweeps}, {sweeps*rounds} rounds, {wall*0.01:.1f} us, {cyc/(sweeps*rounds):.0f} cycles/round, shader clock {cyc/(wall*0.01):.0f} MHz", flush=True)
c.close()
