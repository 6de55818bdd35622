"""cProfile of the host side of a device-mode batch (one batch of B runs, blocking iterations) - diagnostic.
usage: gpu_py_profile_batch.py [B] [dim]"""
import cProfile, pstats, os, sys, io
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import torch
from pcabo.batchrun import BatchedPCABO
from pcabo.bbob import BBOBProblem
torch.set_num_threads(4)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 40
r = BatchedPCABO([BBOBProblem(15, i, dim) for i in range(B)], [15000 + 10 * dim + i for i in range(B)], 10 * dim + 50, 3 * dim,
                 acq_kernel="device", host_threads=2)
r.start()
for _ in range(100): r.iteration()
pr = cProfile.Profile(); pr.enable()
for _ in range(100): r.iteration()
pr.disable(); r.finish()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:9000])
