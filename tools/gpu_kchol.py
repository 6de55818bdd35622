"""K(X,X) + Cholesky micro-benchmark (pcabo.kchol_bench, SURVEY.md 8(d)'s grid) - prints one line per shape.
usage: gpu_kchol.py [batches, e.g. 1,30,120] [reps]"""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
from pcabo import kchol_bench
batches = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 30)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
grid = kchol_bench.run(0, batches, reps=reps)
for g in grid:
    us = g["us"]
    print(f"n={g['n']:5d} k={g['k']:3d} B={g['batch']:4d}: wpca {us['wpca']:7.1f} gram {us['gram']:7.1f} chol {us['cholesky']:7.1f} "
          f"rootinv {us['root_inverse_alpha']:7.1f} us | gram {g['gram_tflops']:6.2f} chol {g['cholesky_tflops']:6.2f} "
          f"rootinv {g['root_inverse_tflops']:6.2f} K+chol {g['kchol_tflops']:6.2f} TF = {100 * g['kchol_frac_of_fp64_peak']:5.2f} % of FP64 peak", flush=True)
print(json.dumps(grid))
