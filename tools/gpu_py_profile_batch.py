"""cProfile of the ONE host thread that interleaves S device-mode batches of B runs (pcabo.batchrun.run_interleaved) - diagnostic.
usage: gpu_py_profile_batch.py [B] [dim] [S]"""
import cProfile, pstats, os, sys, io
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import torch
from pcabo import batchrun
from pcabo.batchrun import BatchedPCABO
from pcabo.bbob import BBOBProblem
torch.set_num_threads(4)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 40
S = int(sys.argv[3]) if len(sys.argv) > 3 else 4
budget = 3 * dim + 220                      # 220 iterations
rs = [BatchedPCABO([BBOBProblem(15, S * i + t, dim) for i in range(B)], [15000 + 10 * dim + S * i + t for i in range(B)], budget, 3 * dim,
                   acq_kernel="device", host_threads=max(1, 8 // S)) for t in range(S)]
for r in rs:
    r.start()
pr = cProfile.Profile(); pr.enable()
batchrun.run_interleaved(rs, started=True)
pr.disable()
for r in rs:
    r.finish()
print(batchrun.LAST_INTERLEAVE_STATS)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:10000])
