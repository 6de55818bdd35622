"""Where does the initial pick (botorch initialize_q_batch) spend its time inside a run?  (diagnostic)"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import numpy as np, torch
from Algorithms import PCA_BO
from pcabo import initializers as _init
from pcabo.bbob import BBOBProblem
seg = np.zeros(6); cnt = [0]
@torch.inference_mode()
def pick(acq_vals, n, eta=1.0):
    t = [time.perf_counter()]
    v = torch.from_numpy(np.ascontiguousarray(acq_vals, dtype=np.float64)); t.append(time.perf_counter())
    std = v.std(dim=0); z = bool(torch.any(std == 0)); t.append(time.perf_counter())
    max_idx = torch.max(v, dim=0)[1]
    eta_z = eta * ((v - v.mean(dim=0)) / std)
    weights = torch.exp(eta_z)
    while bool(torch.isinf(weights).any()):
        eta_z = eta_z * 0.5; weights = torch.exp(eta_z); seg[5] += 1
    t.append(time.perf_counter())
    idcs = torch.multinomial(weights, n); t.append(time.perf_counter())
    if max_idx not in idcs: idcs[-1] = max_idx
    out = idcs.numpy(); t.append(time.perf_counter())
    seg[:5] += np.diff(t); cnt[0] += 1
    return out
_init.initialize_q_batch = pick
torch.set_num_threads(4)
if os.environ.get("PICK_CLOCK_TORCH_CUDA"):
    torch.cuda.set_device(0); torch.cuda.synchronize()
opt = PCA_BO(budget=450, n_DoE=120, var_threshold=0.95, acquisition_function="expected_improvement", random_seed=15400,
             maximization=False, verbose=False, device=0, DoE_parameters={"criterion": "center", "iterations": 1000})
prob = BBOBProblem(15, 0, 40)
opt._start(prob)
for _ in range(10): opt._bo_iteration(prob)
seg[:] = 0; cnt[0] = 0
t0 = time.perf_counter()
for _ in range(250): opt._bo_iteration(prob)
tot = time.perf_counter() - t0
opt._finish()
print(f"iteration {tot/250*1e3:.3f} ms; pick segments (us): from_numpy {seg[0]/cnt[0]*1e6:.1f}, std+any {seg[1]/cnt[0]*1e6:.1f}, "
      f"max/mean/exp/isinf {seg[2]/cnt[0]*1e6:.1f}, multinomial {seg[3]/cnt[0]*1e6:.1f}, contains+numpy {seg[4]/cnt[0]*1e6:.1f}; halvings per call {seg[5]/cnt[0]:.2f}")
