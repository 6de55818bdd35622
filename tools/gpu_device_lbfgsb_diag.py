"""Device-resident L-BFGS-B (csrc/kernels_lbfgsb.hip) - diagnostic: its evaluation against the group kernel's, then one
optimize call in the three modes (host-paced k_acq_group / host-stepped twin / device-resident) on the same state.
usage: gpu_device_lbfgsb_diag.py [n] [d] [B]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np
from pcabo import _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
d = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
q = 512
rng = np.random.default_rng(21)
X = rng.uniform(-5, 5, (B, n, d))
y = rng.normal(size=(B, n)) * 50 + 300
ranks = np.argsort(np.argsort(y, axis=1), axis=1) + 1
noise = rng.normal(0, 1e-8, (B, n, d))
bt = N.Batch(B, max_n=max(n, 64), max_d=d, max_q=q, device_lbfgsb=1)
bt.wpca_gp_condition_begin(X, ranks, noise, y)
res = bt.wpca_results()
boxes = bt.acq_bounds()
raw = [boxes[b][0] + (boxes[b][1] - boxes[b][0]) * rng.uniform(size=(q, res[b]["k"])) for b in range(B)]
best = [float(y[b].min()) for b in range(B)]
vals, status = bt.gp_wait_eval(raw, best)
print("k", [r["k"] for r in res], "status", status.tolist(), flush=True)
order = [np.argsort(-vals[b])[:10] for b in range(B)]
ics = [raw[b][order[b]] for b in range(B)]
# ---- evaluation: device kernel against the group kernel of the same contexts
dv, dg = bt.device_acq_eval(ics, best)
for b in range(B):
    c = bt.ctx[b]
    gv, gg = c.acq_eval(ics[b], best[b], False)
    print("run %d eval: |dval| %.3e (scale %.2e)  |dgrad| %.3e (scale %.2e)" % (b, np.abs(dv[b] - gv).max(), np.abs(gv).max(),
                                                                                np.abs(dg[b] - gg).max(), np.abs(gg).max()), flush=True)
outs = {}
for mode, name in ((0, "group"), (2, "twin"), (1, "device")):
    N.LIB.pcabo_batch_set_option(bt._h, N.OPT_DEVICE_LBFGSB, mode)
    t0 = time.perf_counter()
    o, st = bt.optimize_acqf(ics, boxes, best)
    dt = time.perf_counter() - t0
    outs[name] = o
    print("%-6s %.1f ms status %s" % (name, 1e3 * dt, st.tolist()), flush=True)
    for b in range(B):
        print("   run %d info %s failed %s best val %.6f" % (b, o[b][2].tolist(), o[b][3], o[b][1].max()), flush=True)
for b in range(B):
    t, dv_, g = outs["twin"][b], outs["device"][b], outs["group"][b]
    print("run %d: device vs twin: cand equal %s vals equal %s info equal %s; |dcand| %.3e; twin vs group |dcand| %.3e" % (
        b, np.array_equal(t[0], dv_[0]), np.array_equal(t[1], dv_[1]), np.array_equal(t[2], dv_[2]),
        np.abs(t[0] - dv_[0]).max(), np.abs(t[0] - g[0]).max()), flush=True)
