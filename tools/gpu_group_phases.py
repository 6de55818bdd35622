"""In-kernel phase timeline of k_acq_group (timing build: `make -C para-ortho-pca-bo_amd/csrc timing`, run with
PCABO_LIB=para-ortho-pca-bo_amd/lib/libpcabo_timing.so).  One restart group of 5 queries at (n, k); stamps of the LAST slab
(longest rows) and of the finishing work-group, in microseconds from kernel entry (wall_clock64, 100 MHz)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np
from pcabo import _native as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 449
k = int(sys.argv[2]) if len(sys.argv) > 2 else 36
rng = np.random.default_rng(0)
Z = rng.uniform(0, 1, (n, k)); y = rng.normal(size=n)
c = N.Context(max_n=max(n, 64), max_d=k, max_q=64)
c.gp_condition(y, Z=Z)
c.set_option(N.OPT_GROUP_ACQ, 1)
X = rng.uniform(0, 1, (5, k))
names = ["entry", "xn (PCIe read)", "ks", "v (pass 1)", "w (pass 2)", "ts/tm", "contraction+drain", "finisher starts", "finish done", "flags out"]
for rep in range(3):
    c.acq_eval(X, float(y.min()))
    st = (C.c_ulonglong * 16)()
    assert N.LIB.pcabo_debug_acq_stamps(st) == 0
    t = [int(v) for v in st[:10]]
    print("rep", rep, " ".join("%s=%.2f" % (names[i], (t[i] - t[0]) / 100.0) for i in range(1, 10)))
    t2 = [int(v) for v in st[10:16]]
    if t2[0]:
        print("   contraction detail (wave 0): ts in regs=%.2f first row arrived=%.2f fmas=%.2f lds written=%.2f comp 0 done=%.2f comp 1 done=%.2f"
              % tuple((v - t[0]) / 100.0 for v in t2))
c.close()
