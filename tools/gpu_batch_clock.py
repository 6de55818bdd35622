"""Aggregate rate of B runs advancing in lock-step (pcabo.batchrun) - diagnostic.
usage: gpu_batch_clock.py B [dim] [fid] [sub_batches] [workers per batch, 0 = default] [acq_kernel: group | latency | device] [schedule: threads | interleaved] [CUs of the device optimiser, 0 = all] [1: objectives on the device] [pca | vanilla]"""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # read by the HIP runtime at its first call: a Batch uses a stream per worker thread
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import torch
from pcabo import batchrun
torch.set_num_threads(4)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 40
fid = int(sys.argv[3]) if len(sys.argv) > 3 else 15
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1
W = int(sys.argv[5]) if len(sys.argv) > 5 else 0
AK = sys.argv[6] if len(sys.argv) > 6 else "group"
SCH = sys.argv[7] if len(sys.argv) > 7 else "threads"
out = batchrun.bench_block(0, B, fid, dim, sub_batches=S, workers=W, acq_kernel=AK, schedule=SCH, lbfgsb_cus=int(sys.argv[8]) if len(sys.argv) > 8 else 0,
                           device_objective=bool(int(sys.argv[9])) if len(sys.argv) > 9 else False,
                           algorithm=sys.argv[10] if len(sys.argv) > 10 else "pca")
out["schedule"] = SCH
out["lbfgsb_cus"] = int(sys.argv[8]) if len(sys.argv) > 8 else 0
out["workers"] = W
out["acq_kernel"] = AK
out.pop("best_f")
print(json.dumps(out))
