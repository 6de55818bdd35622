#!/bin/bash
# Device-resident L-BFGS-B (first used in round 3; $1 = round tag, default r04) (k_lbfgsb_group): (1) rocprofv3 --kernel-trace --stats over one 30-run batch of the headline
# cell in device mode; (2) FETCH_SIZE / WRITE_SIZE (separate --pmc passes, kernel trace only) over a shorter batch (8 runs), summed
# per kernel: fabric bytes of k_lbfgsb_group per launch and per L-BFGS-B evaluation against the algorithmic bytes.
# Run on the GPU box from the repo root; writes gpurun_out/<tag>dev/.
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}dev
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $OUT/trace -o batch30dev --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_batch_clock.py 30 40 15 1 0 device > $OUT/batch30_device_under_rocprof.json 2> $OUT/batch30_device_under_rocprof.err
echo "trace: exit $?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_batch_clock.py 8 40 15 1 0 device > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
  echo "$C pass: exit $?"
done
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out = sys.argv[1]
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/gpu_batch_clock.py 8 40 15 1 0 device (8 runs of configs[1]'s "
               "cell, 330 lock-step iterations, n = 120 .. 449); fabric bytes = 2 * FETCH_SIZE KB (gfx950) + WRITE_SIZE KB; per kernel: dispatches, bytes"}
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[name][0] += 1
                acc[name][1] += float(r["Counter_Value"])
    tot[c] = acc
    try:
        res[c + "_run"] = {k: v for k, v in json.loads(open(f"{out}/pmc_{c}.json").read().strip().splitlines()[-1]).items()
                           if k in ("runs", "aggregate_bo_iterations_per_s", "seconds", "bo_iterations")}
    except Exception as e:
        res[c + "_run"] = str(e)
res["kernels"] = {}
for name in sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"])):
    f, w = tot["FETCH_SIZE"].get(name, [0, 0.0]), tot["WRITE_SIZE"].get(name, [0, 0.0])
    res["kernels"][name] = {"dispatches": max(f[0], w[0]), "fetch_bytes": 2 * 1024 * f[1], "write_bytes": 1024 * w[1]}
json.dump(res, open(f"{out}/pmc_device_lbfgsb.json", "w"), indent=1)
k = res["kernels"].get("k_lbfgsb_group")
print(json.dumps({"k_lbfgsb_group": k, "runs": res.get("FETCH_SIZE_run")}))
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
ls $OUT $OUT/trace/* | head -30
