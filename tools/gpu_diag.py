"""Stage-by-stage GPU diagnostic (not a pytest file): prints max errors vs the oracle and call timings."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "para-ortho-pca-bo_amd"), os.path.join(ROOT, "oracle"), ROOT):
    sys.path.insert(0, p)
import pcabo_oracle as O  # noqa: E402
from pcabo import _native as N  # noqa: E402
from pcabo.bbob import BBOBProblem  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def stage(name, fn):
    t = time.perf_counter()
    try:
        out = fn()
        print(f"[{name}] ok {1e3 * (time.perf_counter() - t):.2f} ms", out if out is not None else "", flush=True)
        return True
    except Exception as e:  # noqa: BLE001
        print(f"[{name}] FAILED: {type(e).__name__}: {e}", flush=True)
        return False


def main():
    torch.set_num_threads(8)
    print("devices", N.device_count(), flush=True)
    for d, ndoe, budget, seed in ((10, 30, 150, 15100), (40, 120, 450, 15400)):
        prob = BBOBProblem(15, 0, d)
        o = O.OraclePCABO(budget=budget, n_DoE=ndoe, random_seed=seed, record=True)
        t = time.perf_counter()
        o(prob, d, np.array([-5.0, 5.0]), max_iters=2)
        print(f"== d={d}: oracle 2 iterations {time.perf_counter() - t:.2f} s; timing {o.timing}", flush=True)
        rec = o.records[0]
        wp = rec.wpca
        ctx = N.Context(max_n=budget, max_d=d, max_q=512)
        res = {}

        def s_wpca():
            res.update(ctx.wpca(rec.X, ranks=rec.ranks, noise=rec.noise))
            return {"k": (res["k"], wp.k), "mean": rel(res["data_mean"], wp.data_mean),
                    "pmean": float(np.abs(res["pca_mean"] - wp.pca_mean).max()), "evr": rel(res["evr"], wp.evr),
                    "comps_k": float(np.abs(res["components"][:wp.k] - wp.components[:wp.k]).max()),
                    "comps_all": float(np.abs(res["components"] - wp.components).max()),
                    "Z": rel(res["Z"], wp.Z) if res["k"] == wp.k else None}
        if not stage("wpca", s_wpca):
            continue
        gp = O.ExactGP(wp.Z, rec.f, rec.norm_bounds)
        gp.condition()

        def s_gp():
            ctx.gp_condition(rec.f)
            st = ctx.gp_state()
            K = ctx.gram()
            return {"nb": rel(st["norm_bounds"], rec.norm_bounds), "ymean": st["y_mean"] - gp.y_mean.item(),
                    "ystd": st["y_std"] - gp.y_std.item(), "K": float(np.abs(K - gp.K.numpy()).max()),
                    "L": float(np.abs(st["L"] - gp.L.numpy()).max()), "R": rel(st["R"], gp.Linv.numpy()),
                    "alpha": rel(st["alpha"], gp.alpha.numpy()),
                    "LLt-K": float(np.abs(st["L"] @ st["L"].T - K).max()),
                    "RL-I": float(np.abs(st["R"] @ st["L"] - np.eye(rec.n)).max()),
                    "acqb": rel(ctx.acq_bounds(), rec.acq_bounds)}
        if not stage("gp_condition", s_gp):
            continue
        acq = O.Acquisition(gp, rec.best_f, False)

        def s_acq():
            X = np.vstack([rec.trace.ics, rec.trace.cands])
            ov, og = acq.value_and_grad(X)
            v, g = ctx.acq_eval(X, rec.best_f, False)
            return {"val": float(np.abs(v - ov).max()), "grad": rel(g, og), "vals": v[:3].tolist(), "ovals": ov[:3].tolist()}
        stage("acq_eval(20, grad)", s_acq)

        def s_raw():
            v = ctx.acq_eval(rec.trace.raw_X, rec.best_f, False, grad=False)
            return {"val": float(np.abs(v - rec.trace.raw_vals).max())}
        stage("acq_eval(512)", s_raw)

        def s_opt():
            cand, vals, info, failed = ctx.optimize_acqf(rec.trace.ics, rec.acq_bounds, rec.best_f)
            return {"cand": float(np.abs(cand - rec.trace.cands).max()), "vals": float(np.abs(vals - rec.trace.vals).max()),
                    "info": info.tolist(), "oracle": [(t.nit, t.nfev, t.status) for t in rec.trace.lbfgsb], "failed": failed}
        stage("optimize_acqf", s_opt)
        for rep in range(3):
            stage(f"optimize_acqf rep{rep}", lambda: (ctx.optimize_acqf(rec.trace.ics, rec.acq_bounds, rec.best_f)[2].tolist()))
        for name, fn in (("wpca", lambda: ctx.wpca(rec.X, ranks=rec.ranks, noise=rec.noise, want_Z=False) and None),
                         ("gp_condition", lambda: ctx.gp_condition(rec.f)),
                         ("acq512", lambda: ctx.acq_eval(rec.trace.raw_X, rec.best_f, False, grad=False) is None),
                         ("acq10g", lambda: ctx.acq_eval(rec.trace.ics, rec.best_f, False)[0] is None),
                         ("inverse_map", lambda: ctx.inverse_map(rec.cand_z) is None)):
            fn()
            t = time.perf_counter()
            for _ in range(20):
                fn()
            print(f"   time {name}: {1e3 * (time.perf_counter() - t) / 20:.3f} ms/call", flush=True)
        ctx.set_profiling(True)
        ctx.reset_profile()
        ctx.wpca(rec.X, ranks=rec.ranks, noise=rec.noise, want_Z=False)
        ctx.gp_condition(rec.f)
        ctx.optimize_acqf(rec.trace.ics, rec.acq_bounds, rec.best_f)
        print("   profile", ctx.profile(), flush=True)
        ctx.set_profiling(False)
        ctx.close()


if __name__ == "__main__":
    main()
