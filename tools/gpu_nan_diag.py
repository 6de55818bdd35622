"""Where does the run-ending NaN of a blown-up run come from, and does the oracle meet it too?  (round 3 diagnostic)
f21 / instance 25 / d = 40 (profiles/r02/configs2_configs3_runs.json: stops at n = 412).  The device run is taken to its
failure, the oracle is teacher-forced from that state, and the device's L-BFGS-B problems are re-run from Python through the
C ABI (pcabo_lbfgsb_minimize = csrc/lbfgsb.cpp, f/g = pcabo_acq_eval: the arithmetic of the failing call) so that the point
with the NaN gradient can be looked at from both sides."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("para-ortho-pca-bo_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import pcabo_oracle as O
from pcabo import _native as N
from pcabo.bbob import BBOBProblem
from Algorithms import PCA_BO

torch.set_num_threads(4)
np.set_printoptions(precision=4, linewidth=200)
fid, inst, dim = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (21, 25, 40)))
seed = 1000 * fid + 10 * dim + inst
opt = PCA_BO(budget=10 * dim + 50, n_DoE=3 * dim, random_seed=seed, maximization=False, record_trace=True, acq_kernel="group")
try:
    opt(BBOBProblem(fid, inst, dim))
    print("the device run completed: nothing to diagnose"); sys.exit(0)
except N.PcaboError as e:
    print("device run stopped:", e, "at n =", len(opt.f_evals))
n = len(opt.f_evals)
X, f = np.vstack(opt.x_evals), np.array(opt.f_evals, dtype=float)
tr = opt.trace[-1]
print("max |x| =", np.abs(X).max(), " penalised points:", int((f == 1000.0).sum()), "of", n, " best_f =", tr["best_f"])
orc = O.OraclePCABO(budget=n + 1, n_DoE=n, random_seed=0, maximization=False, record=True)
orc.x_evals = [r.copy() for r in X[:n]]; orc.f_evals = [float(v) for v in f[:n]]; orc._assign_new_best()
np.random.set_state(tr["numpy_state"]); torch.set_rng_state(tr["torch_state"])
try:
    rec = orc.step(BBOBProblem(fid, inst, dim), np.full(dim, -5.0), np.full(dim, 5.0))
    print("oracle step: no exception; k =", rec.k, " retried:", rec.trace.retried, " L-BFGS-B:", [(t.nit, t.nfev, t.message) for t in rec.trace.lbfgsb])
except Exception as e:   # noqa: BLE001
    print("oracle step raised:", repr(e)); rec = None
if rec is None:
    sys.exit(0)
k = rec.k
print("acq box width min/max:", (rec.acq_bounds[1] - rec.acq_bounds[0]).min(), (rec.acq_bounds[1] - rec.acq_bounds[0]).max(),
      " norm box width:", (rec.norm_bounds[1] - rec.norm_bounds[0]))
ctx = N.Context(max_n=n + 1, max_d=dim, max_q=512)
res = ctx.wpca(rec.X, ranks=rec.ranks, noise=rec.noise)
assert res["k"] == k
ctx.gp_condition(rec.f)
st = ctx.gp_state()
print("device y_mean, y_std:", st["y_mean"], st["y_std"], " alpha finite:", np.isfinite(st["alpha"]).all())
gp = O.ExactGP(rec.wpca.Z, rec.f, rec.norm_bounds); gp.condition()
acq = O.Acquisition(gp, rec.best_f, False)
print("ics equal to the device's:", np.array_equal(rec.trace.ics, tr.get("ics")) if "ics" in tr else "device trace has no ics (it failed before)")
ctx.set_option(N.OPT_GROUP_ACQ, 1)
lo, hi = np.tile(rec.acq_bounds[0], 5), np.tile(rec.acq_bounds[1], 5)


found = {}


for g in range(2):
    ics = rec.trace.ics[5 * g:5 * g + 5]
    calls = []

    def fun(x):
        v, gr = ctx.acq_eval(x.reshape(5, k), rec.best_f, False)
        calls.append(x.copy())
        if not np.isfinite(gr).all() or not np.isfinite(v).all():
            found.setdefault(g, len(calls))                # (an exception cannot cross the ctypes callback: stop the run instead)
            return 0.0, np.zeros_like(x)
        return -float(v.sum()), -gr.reshape(-1)

    r = N.lbfgsb_minimize(fun, np.clip(ics.reshape(-1), lo, hi), list(zip(lo, hi)), maxiter=200)
    if g not in found:
        print(f"group {g}: device-surface L-BFGS-B finished without NaN: nit {r['nit']} nfev {r['nfev']}")
        continue
    x = calls[found[g] - 1].reshape(5, k)
    v, gr = ctx.acq_eval(x, rec.best_f, False)
    print(f"group {g}: non-finite value or gradient at evaluation {found[g]}")
    print("    x:", x.ravel(), " box:", rec.acq_bounds.ravel(), " norm bounds:", rec.norm_bounds.ravel())
    print("    device values:", v, " device grad:", gr.ravel())
    print("    oracle values:", acq.value_and_grad(x)[0], " oracle grad:", acq.value_and_grad(x)[1].ravel())
    bad = [j for j in range(5) if not np.isfinite(gr[j]).all() or not np.isfinite(v[j])]
    ov, og = acq.value_and_grad(x)
    ctx.set_option(N.OPT_GROUP_ACQ, 0)
    v2, g2 = ctx.acq_eval(x, rec.best_f, False)
    ctx.set_option(N.OPT_GROUP_ACQ, 1)
    for j in bad:
        xn = (x[j] - rec.norm_bounds[0]) / (rec.norm_bounds[1] - rec.norm_bounds[0])
        with torch.no_grad():
            mean, var = gp.posterior(torch.from_numpy(x[j:j + 1].copy()))
        sigma = float(var.clamp_min(1e-12).sqrt() if hasattr(var, "clamp_min") else np.sqrt(max(float(var), 1e-12)))
        u = (float(mean) - O.round_best_f(rec.best_f)) / sigma
        print(f"  query {j}: device value {v[j]} grad NaNs {int(np.isnan(gr[j]).sum())}/{k} infs {int(np.isinf(gr[j]).sum())}; per-query kernels: value {v2[j]} "
              f"grad NaNs {int(np.isnan(g2[j]).sum())}; oracle value {ov[j]} grad finite {np.isfinite(og[j]).all()} |grad|max {np.abs(og[j]).max():.3e}")
        print(f"    posterior mean {float(mean):.6g} sigma {sigma:.6g} u(min) {-u:.6g}  normalised x range [{xn.min():.4g}, {xn.max():.4g}]  at box edge: "
              f"{int((x[j] <= rec.acq_bounds[0]).sum())} lo / {int((x[j] >= rec.acq_bounds[1]).sum())} hi of {k}")
        print("    oracle grad:", og[j])
        print("    device grad:", gr[j])
ctx.close()
