"""Does a device-mode block of four batches lose its overlap after other blocks have run in the process?  (diagnostic for the
hardware-queue sharing between the batches' streams; prints it/s and the mean 'pca' wait of each block in sequence)
usage: gpu_queue_probe.py [sequence of block names: d4 = 4 x 30 device, g4 = 4 x 30 host-paced, d2, g1, d8 = 4 x 60 device]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
import torch
from pcabo import batchrun
torch.set_num_threads(4)
seq = sys.argv[1:] or ["d4", "g4", "d4"]
for name in seq:
    kind, S = name[0], int(name[1:])
    runs = {1: 30, 2: 60, 4: 120, 8: 240}[S]
    sub = min(S, 4)
    out = batchrun.bench_block(0, runs, 15, 40, sub_batches=sub, acq_kernel="device" if kind == "d" else "group",
                               schedule="interleaved" if kind == "d" else "threads")
    print(name, runs, "runs as", sub, "batches:", round(out["aggregate_bo_iterations_per_s"]), "it/s; pca wait", round(out["host_phase_seconds"]["pca"], 2), "s of", round(out["seconds"], 2), flush=True)
