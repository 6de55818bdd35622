"""Fixed shapes for the PMC passes (one conditioning + one scoring + a few acquisition evaluations per shape):
(n, k) in {(450, 36), (1050, 89)} x batch in {1, 30}.  Run under `rocprofv3 --pmc <counters> --kernel-trace`;
dispatches are told apart by kernel name and grid size (grid z = batch).  usage: gpu_pmc_shapes.py [reps] [batch sizes, e.g. 120]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np
from pcabo import _native as N
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
batches = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 30)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
for n, k in ((450, 36), (1050, 89)):
    for B in batches:
        rng = np.random.default_rng(1000 * n + B)
        X = rng.uniform(1.0 / 12, 11.0 / 12, (B, n, k))
        y = rng.normal(size=(B, n))
        ranks = np.argsort(np.argsort(y, axis=1), axis=1) + 1
        bt = N.Batch(B, max_n=n, max_d=k, max_q=512)
        for r in range(reps):
            bt.wpca_gp_condition_begin(X, ranks, None, y, n_components=k)
            bt.wpca_results()
            boxes = bt.acq_bounds()
            raw = [boxes[b][0] + (boxes[b][1] - boxes[b][0]) * rng.uniform(size=(512, k)) for b in range(B)]
            best = [float(y[b].min()) for b in range(B)]
            vals, st = bt.gp_wait_eval(raw, best)
            assert not st.any()
            ics = [raw[b][:10] for b in range(B)]
            if (n <= 512 and B <= 30) or B == 1:
                bt.optimize_acqf(ics, boxes, best, maxiter=3)          # a few L-BFGS-B rounds through k_acq_group
        if B == 1:                                                   # the single-run latency kernels on the same state
            c = bt.ctx[0]
            c.set_option(N.OPT_GROUP_ACQ, 0)
            for r in range(reps):
                c.acq_eval(ics[0], best[0])
        bt.close()
print("done")
