#!/bin/bash
# Per-dispatch times of the Cholesky kernels at one shape (rocprofv3 kernel trace): which block columns cost what.
# usage: tools/gpu_kchol_trace.sh <round tag> n k B
set -e
R=${1:-r04}; N=${2:-1050}; K=${3:-89}; B=${4:-120}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R/kchol_trace_${N}_${B}
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/gpu_kchol_shape.py $N $K $B 2 > $OUT/run.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    if "chol" not in name and "gram" not in name and "trinv" not in name:
        continue
    key = (name[:40], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Z"]))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
with open(out + "/per_dispatch.txt", "w") as o:
    for (name, gx, gz), (cnt, us) in agg.items():
        o.write(f"{name:40s} groups_x {gx:5d} z {gz:4d} launches {cnt:4d} mean_us {us / cnt:9.2f}\n")
print(open(out + "/per_dispatch.txt").read())
PY
