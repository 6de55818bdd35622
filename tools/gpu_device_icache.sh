#!/bin/bash
# Instruction-cache behaviour and wave-cycle buckets of k_lbfgsb_group (one optimize call of 30 runs at n = 449, three dispatches):
# three --pmc passes (kernel trace only; the third counts the vector memory instructions).  $1 = round tag (default r04); writes gpurun_out/<tag>dev/icache.json.
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}dev
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace -d $OUT/pmc_ic -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_device_lbfgsb_phases.py 449 40 30 > $OUT/pmc_ic.log 2> $OUT/pmc_ic.err
echo "icache pass: exit $?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $OUT/pmc_sq -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_device_lbfgsb_phases.py 449 40 30 > $OUT/pmc_sq.log 2> $OUT/pmc_sq.err
echo "sq pass: exit $?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc_vmem -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_device_lbfgsb_phases.py 449 40 30 > $OUT/pmc_vmem.log 2> $OUT/pmc_vmem.err
echo "vmem pass: exit $?"
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out = sys.argv[1]
res = {"note": "rocprofv3 --pmc over tools/gpu_device_lbfgsb_phases.py 449 40 30: k_lbfgsb_group, 60 work-groups of 16 waves, 3 dispatches summed"}
for tag in ("ic", "sq", "vmem"):
    acc = defaultdict(float); disp = set()
    for f in glob.glob(f"{out}/pmc_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("k_lbfgsb_group"):
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r.get("Dispatch_Id"))
    res[tag] = dict(acc); res[tag + "_dispatches"] = len(disp)
ic = res["ic"]
if ic.get("SQC_ICACHE_REQ"):
    res["icache_hit_rate"] = ic.get("SQC_ICACHE_HITS", 0.0) / ic["SQC_ICACHE_REQ"]
sq = res["sq"]
if sq.get("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        res[k + "_share"] = sq.get(k, 0.0) / sq["SQ_WAVE_CYCLES"]
# evaluations of the three dispatches, from the tool's own output: "evaluations per group: mean M" x 60 groups per call
import re
ev = 0.0
try:
    for line in open(f"{out}/pmc_vmem.log"):
        m = re.search(r"evaluations per group: mean ([0-9.]+)", line)
        if m:
            ev += float(m.group(1)) * 60
except Exception:
    pass
if ev and res["vmem"].get("SQ_INSTS_VMEM_RD"):
    res["group_evaluations"] = ev
    res["vmem_read_instructions_per_group_evaluation"] = res["vmem"]["SQ_INSTS_VMEM_RD"] / ev
    res["vmem_write_instructions_per_group_evaluation"] = res["vmem"].get("SQ_INSTS_VMEM_WR", 0.0) / ev
json.dump(res, open(f"{out}/icache.json", "w"), indent=1)
print(json.dumps(res))
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
