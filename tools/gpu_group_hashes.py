"""sha256 of value + gradient from the throughput kernel k_acq_group (PCABO_OPT_GROUP_ACQ) on seeded GP states and query
sets, and of one whole pcabo_optimize_acqf in that mode.  Phases of the kernel were moved to the matrix cores under the
promise "same bits" (v_mfma_f64_16x16x4 is an ascending fma chain, profiles/tools/mfma_f64_order.hip):
tests/golden/acq_group_hashes.json holds the bits of the VALU version.
    python tools/gpu_group_hashes.py            # print
    python tools/gpu_group_hashes.py --write    # regenerate (only after an INTENDED change of arithmetic)"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np

# (new cases go to the END: one generator runs through all of them)
CASES = [(120, 12), (200, 20), (256, 30), (320, 33), (449, 36), (450, 36), (512, 40), (700, 40), (1050, 89),
         (512, 89), (449, 85)]      # round 3: k_acq_group<2> with more than 64 KB of dynamic LDS (k >= 83 at NP = 512)
GOLDEN = os.path.join(ROOT, "tests", "golden", "acq_group_hashes.json")


def _h(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def compute() -> dict:
    from pcabo import _native as N
    out = {}
    rng = np.random.default_rng(11)
    for n, k in CASES:
        Z = rng.uniform(0, 1, (n, k)); y = rng.normal(size=n)
        c = N.Context(max_n=max(n, 64), max_d=max(k, 2), max_q=64)
        c.set_option(N.OPT_GROUP_ACQ, 1)
        c.gp_condition(y, Z=Z)
        best = float(y.min())
        rec = {}
        for q in (10, 7, 3):
            Xq = rng.uniform(0.05, 0.95, (q, k))
            val, g = c.acq_eval(Xq, best, False, N.ACQ_LOG_EI, grad=True)
            assert np.isfinite(val).all() and np.isfinite(g).all()
            rec[f"q{q}"] = {"val": _h(val), "grad": _h(g)}
        val, g = c.acq_eval(rng.uniform(0.05, 0.95, (5, k)), best, False, N.ACQ_PI, grad=True)
        rec["pi_q5"] = {"val": _h(val), "grad": _h(g)}
        if n <= 450:
            ics = rng.uniform(0.1, 0.9, (10, k))
            bounds = np.vstack([np.zeros(k), np.ones(k)])
            cand, vals, info, failed = c.optimize_acqf(ics, bounds, best, False, N.ACQ_LOG_EI, batch_limit=5, maxiter=60)
            rec["optimize"] = {"cand": _h(cand), "vals": _h(vals), "info": _h(info)}
        out[f"{n},{k}"] = rec
        c.close()
    return out


if __name__ == "__main__":
    res = compute()
    if "--write" in sys.argv:
        out = sys.argv[sys.argv.index("--write") + 1] if len(sys.argv) > sys.argv.index("--write") + 1 else GOLDEN
        with open(out, "w") as f:
            json.dump({"_comment": "tools/gpu_group_hashes.py --write on an MI355X (round 2, VALU v / w phases of k_acq_group; the last "
                                   "two cases added in round 3)", "cases": res}, f, indent=1)
        print("written", out)
    else:
        print(json.dumps(res, indent=1))
