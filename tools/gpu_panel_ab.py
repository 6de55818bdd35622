"""A/B of the two panel kernels (PCABO_PANEL_VALU=1: k_chol_panel_w, default: k_chol_panel_m): the Cholesky factors, root
inverses and alphas of a few conditionings must be equal BIT FOR BIT; prints HIP-event time of the Cholesky group.
usage: gpu_panel_ab.py            (spawns itself twice)"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np

CASES = [(64, 3), (100, 7), (130, 10), (450, 36), (449, 20), (1050, 89), (700, 40)]


def child():
    from pcabo import _native as N
    out = {}
    rng = np.random.default_rng(7)
    for n, k in CASES:
        Z = rng.uniform(0, 1, (n, k)); y = rng.normal(size=n)
        c = N.Context(max_n=max(n, 64), max_d=max(k, 2), max_q=64)
        c.set_profiling(True)
        for _ in range(3):
            c.gp_condition(y, Z=Z)
        st = c.gp_state()
        prof = c.profile()
        out[f"{n},{k}"] = {"L": hashlib.sha256(st["L"].tobytes()).hexdigest(), "R": hashlib.sha256(st["R"].tobytes()).hexdigest(),
                           "alpha": hashlib.sha256(st["alpha"].tobytes()).hexdigest(), "chol_us": round(1e3 * prof["cholesky"]["ms"] / max(1, prof["cholesky"]["launches"]), 2)}
        c.close()
    # a matrix that is not positive definite at first: duplicate points, the jitter retries must behave the same
    Z = rng.uniform(0, 1, (200, 5)); Z[100:] = Z[:100]; y = rng.normal(size=200)
    c = N.Context(max_n=200, max_d=5, max_q=64)
    try:
        c.gp_condition(y, Z=Z, noise=0.0)
        out["dup"] = {"L": hashlib.sha256(c.gp_state()["L"].tobytes()).hexdigest()}
    except Exception as e:      # noqa: BLE001
        out["dup"] = {"error": str(e)[:120]}
    c.close()
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        res = {}
        for tag, env in (("valu", {"PCABO_PANEL_VALU": "1"}), ("mfma", {})):
            e = dict(os.environ); e.update(env)
            p = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
            if p.returncode != 0:
                print(tag, "FAILED", p.stderr[-2000:]); sys.exit(1)
            res[tag] = json.loads(p.stdout.strip().splitlines()[-1])
        ok = True
        for key in res["valu"]:
            a, b = res["valu"][key], res["mfma"][key]
            same = all(a[f] == b[f] for f in a if f != "chol_us")
            ok &= same
            print(key, "identical" if same else "DIFFERENT", "chol us valu/mfma:", a.get("chol_us"), b.get("chol_us"), a.get("error", ""))
        print("ALL IDENTICAL" if ok else "MISMATCH")
        sys.exit(0 if ok else 2)
