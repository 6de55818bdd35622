"""Micro-timing of the acquisition launch at a few (n, k) sizes (diagnostic; variants via PCABO_ACQ_VARIANT)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
from pcabo import _native as N
rng = np.random.default_rng(0)
for n, k in ((120, 31), (250, 33), (449, 36)):
    Z = rng.uniform(-1, 1, size=(n, k)); y = rng.normal(size=n)
    c = N.Context(max_n=450, max_d=40, max_q=512)
    c.gp_condition(y, Z=Z)
    Xq = rng.uniform(-1, 1, size=(10, k)); Xr = rng.uniform(-1, 1, size=(512, k))
    for name, fn in (("q10 grad", lambda: c.acq_eval(Xq, 0.0)), ("q5 grad", lambda: c.acq_eval(Xq[:5], 0.0)), ("q512 val", lambda: c.acq_eval(Xr, 0.0, grad=False))):
        for _ in range(20): fn()
        t = time.perf_counter()
        for _ in range(200): fn()
        print(f"variant={os.environ.get('PCABO_ACQ_VARIANT','0')} n={n} k={k} {name}: {(time.perf_counter()-t)/200*1e6:.1f} us/call", flush=True)
    c.close()
