"""Find which (function, instance, iteration) of the d=40 cells produces a NaN acquisition gradient (diagnostic)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np, torch
from pcabo.batchrun import BatchedPCABO
from pcabo.bbob import BBOBProblem
from pcabo import _native as N
torch.set_num_threads(4)
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 40
fids = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(15, 25))
for fid in fids:
    r = BatchedPCABO([BBOBProblem(fid, i, dim) for i in range(30)], [1000 * fid + 10 * dim + i for i in range(30)], 10 * dim + 50, 3 * dim)
    r.start()
    try:
        while r.n < r.budget:
            r.iteration()
        print(fid, "ok", flush=True)
    except N.PcaboError as e:
        n = r.n
        print(fid, "FAILED at n =", n, e, flush=True)
        bad = [b for b in range(30) if not np.isfinite(r.f_evals[b]).all()]
        print("  runs with non-finite f:", bad)
        for b in range(30):
            f = np.array(r.f_evals[b]); X = np.vstack(r.x_evals[b])
            if b in (29,) or b in bad:
                print("  run", b, "f min/max", f.min(), f.max(), "|x| max", np.abs(X).max(), "k", int(r._batch.k[b]), "best", r.current_best[b])
                c = r._batch.ctx[b]
                try:
                    st = c.gp_state()
                    print("    y_mean, y_std", st["y_mean"], st["y_std"], "alpha finite", np.isfinite(st["alpha"]).all(), "R finite", np.isfinite(st["R"]).all(),
                          "norm_bounds finite", np.isfinite(st["norm_bounds"]).all(), "range min", (st["norm_bounds"][1] - st["norm_bounds"][0]).min())
                    box = c.acq_bounds()
                    Xq = box[0] + (box[1] - box[0]) * np.random.default_rng(0).uniform(size=(10, c.k))
                    v, g = c.acq_eval(Xq, r.current_best[b])
                    print("    sample values", v[:4], "grad finite", np.isfinite(g).all(), "box width min/max", (box[1]-box[0]).min(), (box[1]-box[0]).max())
                except Exception as ee:
                    print("    state error", ee)
    finally:
        r.finish()
