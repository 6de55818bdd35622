"""In-kernel clocks of the device-resident L-BFGS-B (timing build: `make -C para-ortho-pca-bo_amd/csrc timing`, run with
PCABO_LIB=para-ortho-pca-bo_amd/lib/libpcabo_timing.so): microseconds per call of every step routine and of the evaluation's
phases, work-group 0 of one optimize call on a synthetic state.  usage: gpu_device_lbfgsb_phases.py [n] [d] [B]"""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np
from pcabo import _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 449
d = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 30
q = 512
rng = np.random.default_rng(3)
X = rng.uniform(-5, 5, (B, n, d))
y = (X ** 2).sum(axis=2) + 10 * np.cos(X).sum(axis=2) + rng.normal(size=(B, n))
ranks = np.argsort(np.argsort(y, axis=1), axis=1) + 1
noise = rng.normal(0, 1e-8, (B, n, d))
bt = N.Batch(B, max_n=max(n, 64), max_d=d, max_q=q, device_lbfgsb=1)
bt.wpca_gp_condition_begin(X, ranks, noise, y)
res = bt.wpca_results()
boxes = bt.acq_bounds()
raw = [boxes[b][0] + (boxes[b][1] - boxes[b][0]) * rng.uniform(size=(q, res[b]["k"])) for b in range(B)]
best = [float(y[b].min()) for b in range(B)]
vals, status = bt.gp_wait_eval(raw, best)
ics = [raw[b][np.argsort(-vals[b])[:10]] for b in range(B)]
names = {0: "cauchy", 1: "freev", 2: "formk", 3: "cmprlb", 4: "subsm", 5: "lnsrlb", 6: "matupd", 7: "formt",
         8: "eval: xn", 9: "eval: ks", 10: "eval: pass 1", 11: "eval: combine v", 12: "eval: scalar + pass 2", 13: "eval: u",
         14: "eval: contraction", 16: "step (advance, all of it)", 17: "evaluation (all of it)",
         18: "formk: shift + fill", 19: "formk: accum", 20: "formk: new column", 21: "formk: corrections", 22: "formk: assemble WN",
         23: "formk: dpofa 1", 24: "formk: solves", 25: "formk: products", 26: "formk: dpofa 2",
         32: "cauchy: classify", 33: "cauchy: f1 + accum + copy", 34: "cauchy: wait for formt", 35: "cauchy: bmv + ddot", 36: "cauchy: breakpoint loop", 37: "lnsrlb: head (stpmx scan, copies / state load)", 38: "lnsrlb: g'd", 39: "lnsrlb: dcsrch + trial point", 40: "cauchy: breakpoints crossed (count)", 41: "step: head of an iteration (tests, r = g - r, r'r)", 42: "step: posts in front of cauchy", 43: "step: d = z - x", 44: "step: tail of an accepted search (projgr)", 46: "kernel: absorb + barrier",
         27: "subsm: scatter", 28: "subsm: accum", 29: "subsm: solves", 30: "subsm: update full", 31: "subsm: project"}
have = hasattr(N.LIB, "pcabo_debug_lb_ticks")
for rep in range(3):
    if have:
        t, c = (C.c_ulonglong * 64)(), (C.c_ulonglong * 64)()
        N.LIB.pcabo_debug_lb_ticks(t, c, 1)
    t0 = time.perf_counter()
    o, st = bt.optimize_acqf(ics, boxes, best)
    dt = time.perf_counter() - t0
    info = np.array([o[b][2] for b in range(B)])
    print("call %d: %.2f ms; k %s; evaluations per group: mean %.1f max %d (group 0 of run 0: %d iterations, %d evaluations)" % (
        rep, 1e3 * dt, sorted({int(r["k"]) for r in res}), info[:, :, 1].mean(), info[:, :, 1].max(), info[0, 0, 0], info[0, 0, 1]), flush=True)
    if have:
        N.LIB.pcabo_debug_lb_ticks(t, c, 0)
        for i in sorted(names):
            if c[i]:
                print("   %-28s %6d calls  %8.2f us per call  %9.1f us in total" % (names[i], c[i], t[i] / 100.0 / c[i], t[i] / 100.0))
