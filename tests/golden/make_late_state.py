"""Generates tests/golden/late_state_d40.npz: late-phase states of the headline configuration (BBOB f15, d = 40, instance 0,
seed 15400 - BASELINE.json configs[1]) from a FREE-RUNNING CPU ORACLE run (no GPU, no reference import): the design matrix
and objective values of the first 420 evaluations plus the numpy / torch generator states in front of the iterations at
n = 320, 384 and 420.  tests/test_lbfgsb_divergence.py teacher-forces the oracle from these states.
    python tests/golden/make_late_state.py          (about two minutes on one core)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "para-ortho-pca-bo_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import pcabo_oracle as O                      # noqa: E402
from pcabo.bbob import BBOBProblem            # noqa: E402

NS = (320, 384, 420)
torch.set_num_threads(1)
prob = BBOBProblem(15, 0, 40)
o = O.OraclePCABO(budget=450, n_DoE=120, random_seed=15400)
lb, ub = np.full(40, -5.0), np.full(40, 5.0)
o.initial_design(prob, 40, lb, ub)
states = {}
while len(o.f_evals) < max(NS) + 1:
    n = len(o.f_evals)
    if n in NS:
        st = np.random.get_state()
        states[n] = (st[1].copy(), int(st[2]), int(st[3]), float(st[4]), torch.get_rng_state().numpy().copy())
    o.step(prob, lb, ub)
    if n % 20 == 0:
        print(n, o.current_best, flush=True)
out = {"X": np.vstack(o.x_evals)[: max(NS)], "f": np.array(o.f_evals[: max(NS)]), "ns": np.array(NS)}
for n, (key, pos, has_gauss, gauss, tstate) in states.items():
    out[f"np_key_{n}"], out[f"np_meta_{n}"], out[f"torch_{n}"] = key, np.array([pos, has_gauss, gauss]), tstate
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "late_state_d40.npz"), **out)
print("written")
