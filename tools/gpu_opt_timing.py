"""Where does an optimize_acqf call spend its time?  (diagnostic, PCABO_TRACE_OPT=1 prints host/eval split)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
from pcabo import _native as N
rng = np.random.default_rng(0)
for n, k in ((120, 31), (250, 33), (449, 36)):
    Z = rng.uniform(-1, 1, size=(n, k)); y = rng.normal(size=n) * 300 + 2000
    c = N.Context(max_n=450, max_d=40, max_q=512)
    c.gp_condition(y, Z=Z)
    b = c.acq_bounds()
    ics = rng.uniform(b[0], b[1], size=(10, k))
    for rep in range(3):
        t = time.perf_counter()
        cand, vals, info, failed = c.optimize_acqf(ics, b, float(y.min()))
        print(f"n={n} optimize {1e3*(time.perf_counter()-t):.2f} ms info {info.tolist()}", flush=True)
    c.close()
