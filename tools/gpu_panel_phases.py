"""In-kernel timeline of k_chol_panel_m (timing build): wave 0 of block 0 of the LAST panel launch of a conditioning at
n = 450 (8 panels; the last launch has one block).  Stamps in microseconds from kernel entry."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np
from pcabo import _native as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 450
rng = np.random.default_rng(0)
Z = rng.uniform(0, 1, (n, 20)); y = rng.normal(size=n)
c = N.Context(max_n=max(n, 64), max_d=20, max_q=64)
names = ["entry", "tile in registers", "sub0 start", "sub0 pivots done", "sub0 published", "sub1 start", "sub1 pivots", "sub1 published",
         "sub2 start", "sub2 pivots", "sub2 published", "sub3 start", "sub3 pivots", "sub3 published", "factor done", "stored"]
for rep in range(3):
    c.gp_condition(y, Z=Z)
    st = (C.c_ulonglong * 16)()
    assert N.LIB.pcabo_debug_panel_stamps(st) == 0
    t = [int(v) for v in st]
    print("rep", rep, " ".join("%s=%.2f" % (names[i], (t[i] - t[0]) / 100.0) for i in range(1, 16)))
c.close()
