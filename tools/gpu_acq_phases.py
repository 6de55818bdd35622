"""In-kernel phase timeline of k_acq_fused (diagnostic; needs `make -C para-ortho-pca-bo_amd/csrc timing`).
Stamps are wall_clock64() ticks (100 MHz) of the middle slab group of query 0 and of the group that finishes it."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PCABO_LIB"] = os.path.join(ROOT, "para-ortho-pca-bo_amd", "lib", "libpcabo_timing.so")
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
from pcabo import _native as N
names = ["xn", "ks", "v", "vv/mu", "w", "contract", "drain+barrier", "ticket->finisher", "finish+publish", "host drain"]
rng = np.random.default_rng(0)
for n, k in ((120, 31), (250, 33), (449, 36), (150, 27), (250, 17), (350, 10), (430, 8)):
    Z = rng.uniform(-1, 1, size=(n, k)); y = rng.normal(size=n)
    c = N.Context(max_n=450, max_d=40, max_q=512)
    c.gp_condition(y, Z=Z)
    Xq = rng.uniform(-1, 1, size=(10, k))
    acc = np.zeros(10); cnt = 0
    for it in range(60):
        c.acq_eval(Xq, 0.0)
        time.sleep(0.0005)
        st = (C.c_ulonglong * 16)()
        assert N.LIB.pcabo_debug_acq_stamps(st) == 0
        t = np.array(list(st)[:11], dtype=np.float64)
        if it >= 10:
            acc += np.diff(t) * 0.01; cnt += 1          # us
            fin = np.array([st[8], st[11], st[12], st[13], st[9]], dtype=np.float64)
            facc = (facc if it > 10 else np.zeros(4)) + np.diff(fin) * 0.01
    t0 = time.perf_counter()
    for _ in range(200): c.acq_eval(Xq, 0.0)
    wall = (time.perf_counter() - t0) / 200 * 1e6
    print(f"n={n} k={k} q=10: host round trip {wall:.1f} us; in-kernel total {acc.sum()/cnt:.1f} us: " +
          ", ".join(f"{nm} {v/cnt:.2f}" for nm, v in zip(names, acc)) +
          " | finish split: partial loads+sums %.2f, scalar chain %.2f, wait for waves 1-3 %.2f, gradient+stores+flag %.2f" % tuple(facc / cnt), flush=True)
    c.close()
