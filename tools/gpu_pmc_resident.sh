#!/bin/bash
# Fabric traffic of the RESIDENT acquisition kernel (k_acq_fast<SLAB,NB,true>: one dispatch per optimize call, what bench.py's
# timed region executes): FETCH_SIZE and WRITE_SIZE in separate passes over a whole run of configs[1] (no tracing domain
# besides the kernel trace), summed per instantiation; divided by the L-BFGS-B rounds the run reports this is the traffic per
# round.  Run on the GPU box from the repo root; writes gpurun_out/<tag>/pmc_resident.json.
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_res_$C -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 0 --batch 0 --no-kchol-grid --no-roofline --no-cpu-baseline > $OUT/pmc_res_$C.json 2> $OUT/pmc_res_$C.err
  echo "$C pass: exit $?"
done
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out = sys.argv[1]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{out}/pmc_res_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if r["Counter_Name"] == c and name.startswith("k_acq_fast"):
                acc[name][0] += 1
                acc[name][1] += float(r["Counter_Value"])
    tot[c] = acc
    line = json.loads(open(f"{out}/pmc_res_{c}.json").read().strip().splitlines()[-1])
    tot[c + "_rounds"] = line["host_phase_seconds"]["lbfgsb_rounds"]
    tot[c + "_value"] = line["value"]
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 6 --warmup 0: every dispatch of the run; "
               "fabric bytes = 2 * FETCH_SIZE KB (gfx950) + WRITE_SIZE KB; resident = k_acq_fast<..., true>, one dispatch per optimize call",
       "lbfgsb_rounds": tot["FETCH_SIZE_rounds"], "bo_iterations_per_s_under_pmc": [tot["FETCH_SIZE_value"], tot["WRITE_SIZE_value"]], "kernels": {}}
fb = wb = nd = 0.0
for name in sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"])):
    f, w = tot["FETCH_SIZE"].get(name, [0, 0.0]), tot["WRITE_SIZE"].get(name, [0, 0.0])
    res["kernels"][name] = {"dispatches": f[0], "fetch_bytes": 2 * 1024 * f[1], "write_bytes": 1024 * w[1]}
    if ", true>" in name:
        fb += 2 * 1024 * f[1]; wb += 1024 * w[1]; nd += f[0]
res["resident_total_bytes"] = fb + wb
res["resident_dispatches"] = nd
res["resident_bytes_per_round"] = (fb + wb) / max(1, tot["FETCH_SIZE_rounds"])
res["resident_bytes_per_dispatch"] = (fb + wb) / max(1.0, nd)
json.dump(res, open(f"{out}/pmc_resident.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
