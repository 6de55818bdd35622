"""sha256 of the Cholesky factor L, the root inverse R = L^-1 and alpha after pcabo_gp_condition on seeded inputs.
The conditioning kernels are deterministic, and several of them were rewritten under the promise "same bits" (left-looking
Cholesky, matrix-core trailing updates in the panel kernel - profiles/r02/panel_ab.txt): tests/golden/gp_factor_hashes.json
pins the bits so that the next rewrite is held to the same promise (tests/test_gpu_parity.py).
    python tools/gpu_factor_hashes.py            # print
    python tools/gpu_factor_hashes.py --write    # regenerate the golden file (only after an INTENDED change of arithmetic)"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np

CASES = [(64, 3), (100, 7), (130, 10), (450, 36), (449, 20), (1050, 89), (700, 40)]
GOLDEN = os.path.join(ROOT, "tests", "golden", "gp_factor_hashes.json")


def compute() -> dict:
    from pcabo import _native as N
    out = {}
    rng = np.random.default_rng(7)
    for n, k in CASES:
        Z = rng.uniform(0, 1, (n, k)); y = rng.normal(size=n)
        c = N.Context(max_n=max(n, 64), max_d=max(k, 2), max_q=64)
        c.gp_condition(y, Z=Z)
        st = c.gp_state()
        out[f"{n},{k}"] = {f: hashlib.sha256(np.ascontiguousarray(st[f]).tobytes()).hexdigest() for f in ("L", "R", "alpha")}
        c.close()
    # not positive definite at first (duplicate points, no noise): the jitter retries are part of the pinned behaviour
    Z = rng.uniform(0, 1, (200, 5)); Z[100:] = Z[:100]; y = rng.normal(size=200)
    c = N.Context(max_n=200, max_d=5, max_q=64)
    c.gp_condition(y, Z=Z, noise=0.0)
    out["duplicates,200,5"] = {"L": hashlib.sha256(np.ascontiguousarray(c.gp_state()["L"]).tobytes()).hexdigest()}
    c.close()
    return out


if __name__ == "__main__":
    res = compute()
    if "--write" in sys.argv:
        with open(GOLDEN, "w") as f:
            json.dump({"_comment": "tools/gpu_factor_hashes.py --write on an MI355X (round 2; identical for k_chol_panel_w and "
                                   "k_chol_panel_m, profiles/r02/panel_ab.txt)", "cases": res}, f, indent=1)
        print("written", GOLDEN)
    else:
        print(json.dumps(res, indent=1))
