"""Soak: R repetitions of 3 concurrent processes (instances 0..2 of the headline run) on one GPU; every repetition must
reproduce the same per-instance digest (diagnostic, prints a summary)."""
import hashlib, os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, hashlib
import numpy as np
sys.path.insert(0, os.path.join(%r, "para-ortho-pca-bo_amd"))
from Algorithms import PCA_BO
from pcabo.bbob import BBOBProblem
inst = int(sys.argv[1])
for rep in range(int(sys.argv[2])):
    opt = PCA_BO(budget=450, n_DoE=120, random_seed=15400 + inst, maximization=False)
    opt(BBOBProblem(15, inst, 40))
    print("DIGEST", inst, hashlib.md5(np.array(opt.f_evals).tobytes() + np.vstack(opt.x_evals).tobytes()).hexdigest(), flush=True)
''' % root
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
t0 = time.time()
procs = [subprocess.Popen([sys.executable, "-c", code, str(i), str(reps)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(3)]
digests = {}
for p in procs:
    so, se = p.communicate(timeout=1500)
    assert p.returncode == 0, se[-2000:]
    for l in so.splitlines():
        if l.startswith("DIGEST"):
            _, inst, d = l.split(); digests.setdefault(inst, set()).add(d)
print("wall %.1f s;" % (time.time() - t0), {k: len(v) for k, v in digests.items()}, "distinct digests per instance (1 = deterministic)")
assert all(len(v) == 1 for v in digests.values())
print("OK")
