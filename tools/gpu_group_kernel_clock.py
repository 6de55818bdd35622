"""Time of the batched L-BFGS-B rounds in isolation: B runs conditioned on random data of shape (n, k), then repeated
optimize calls from fixed initial conditions (diagnostic; modes via PCABO_BATCH_ACQ=group|slab, PCABO_BATCH_THREADS)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np
from pcabo import _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 449
d = int(sys.argv[3]) if len(sys.argv) > 3 else 36
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
rng = np.random.default_rng(1)
X = rng.uniform(-5, 5, (B, n, d))
y = np.sum(X ** 2, axis=2) + rng.normal(size=(B, n))
ranks = np.argsort(np.argsort(y, axis=1), axis=1) + 1
bt = N.Batch(B, max_n=max(n, 64), max_d=d, max_q=512)
bt.wpca_gp_condition_begin(X, ranks, None, y, n_components=d)
res = bt.wpca_results()
boxes = bt.acq_bounds()
raw = [boxes[b][0] + (boxes[b][1] - boxes[b][0]) * rng.uniform(size=(512, res[b]["k"])) for b in range(B)]
best = [float(y[b].min()) for b in range(B)]
t0 = time.perf_counter(); vals, st = bt.gp_wait_eval(raw, best); t_score = time.perf_counter() - t0
ics = [raw[b][np.argsort(-vals[b])[:10]] for b in range(B)]
outs, st = bt.optimize_acqf(ics, boxes, best)         # warm
t0 = time.perf_counter()
for _ in range(reps):
    outs, st = bt.optimize_acqf(ics, boxes, best)
dt = (time.perf_counter() - t0) / reps
rounds = max(int(o[2][:, 1].max()) for o in outs)
evals = sum(int(o[2][:, 1].sum()) for o in outs)
print(json.dumps({"B": B, "n": n, "k": d, "mode": os.environ.get("PCABO_BATCH_ACQ", "group"), "threads": os.environ.get("PCABO_BATCH_THREADS"),
                  "optimize_ms": 1e3 * dt, "max_rounds": rounds, "group_evals": evals, "us_per_group_eval": 1e6 * dt / evals,
                  "us_per_round": 1e6 * dt / rounds, "score_ms": 1e3 * t_score}))
bt.close()
