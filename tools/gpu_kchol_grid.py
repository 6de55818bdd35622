"""SURVEY 8(d) micro-benchmark grid for K(X,X) + Cholesky (pcabo.kchol_bench); prints one JSON line per shape.
usage: gpu_kchol_grid.py [batches e.g. 1,30,120]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "para-ortho-pca-bo_amd"))
from pcabo import kchol_bench
batches = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 30)
for row in kchol_bench.run(0, batches):
    print(json.dumps(row))
