"""T sub-batches of 30/T runs each, every sub-batch on its own Python thread of ONE process (the C calls release the
interpreter lock): aggregate rate.  usage: gpu_batch_threads.py T [runs] [dim] [fid]   (PCABO_BATCH_THREADS = gang threads per sub-batch)"""
import json, os, sys, threading
from time import perf_counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import torch
from pcabo import batchrun
from pcabo.bbob import BBOBProblem
torch.set_num_threads(1)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 40
fid = int(sys.argv[4]) if len(sys.argv) > 4 else 15
budget, n_doe = 10 * dim + 50, 3 * dim
subs = []
for t in range(T):
    inst = list(range(t, R, T))
    r = batchrun.BatchedPCABO([BBOBProblem(fid, i, dim) for i in inst], [1000 * fid + 10 * dim + i for i in inst], budget, n_doe,
                              host_threads=max(1, 8 // T))
    r.start()
    subs.append(r)
torch.cuda.synchronize()
bar = threading.Barrier(T + 1)
def drive(r):
    bar.wait()
    while r.n < budget:
        r.iteration()
th = [threading.Thread(target=drive, args=(r,)) for r in subs]
for t in th: t.start()
bar.wait()
t0 = perf_counter()
for t in th: t.join()
torch.cuda.synchronize()
dt = perf_counter() - t0
iters = sum(len(f) - n_doe for r in subs for f in r.f_evals)
for r in subs: r.finish()
print(json.dumps({"sub_batches": T, "runs": R, "aggregate_bo_iterations_per_s": iters / dt, "seconds": dt,
                  "gang_threads": os.environ.get("PCABO_BATCH_THREADS"), "phases": [dict(r.timing) for r in subs]}))
