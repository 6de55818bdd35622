"""One shape of the K(X,X) + Cholesky micro-benchmark (pcabo/kchol_bench.py) - for a kernel trace of that shape alone:
rocprofv3 --kernel-trace --stats -- python3 tools/gpu_kchol_shape.py n k B [reps].  Prints the event times of the phases."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from pcabo import kchol_bench
n, k, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
print(json.dumps(kchol_bench.run(0, (B,), grid=((n, k),), reps=reps)))
