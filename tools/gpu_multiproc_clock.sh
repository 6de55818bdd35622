#!/bin/bash
# P concurrent processes, each running gpu_batch_clock.py with the given arguments (one GPU): aggregate = sum of the processes' rates
# over the time they overlap (they start within a second of each other; each line reports its own clock).
# usage: gpu_multiproc_clock.sh P B dim fid S W acq_kernel out_prefix
P=$1; shift; OUT=${7:-gpurun_out/mp}
pids=()
for p in $(seq 1 $P); do
  python tools/gpu_batch_clock.py $1 $2 $3 $4 $5 $6 > ${OUT}_$p.json 2> ${OUT}_$p.err &
  pids+=($!)
done
rc=0
for pid in "${pids[@]}"; do wait $pid || rc=1; done
python - "$OUT" "$P" <<'PY'
import json, sys
out, P = sys.argv[1], int(sys.argv[2])
rows = [json.load(open(f"{out}_{p}.json")) for p in range(1, P + 1)]
print(json.dumps({"processes": P, "runs_per_process": rows[0]["runs"], "sub_batches": rows[0]["sub_batches"], "acq_kernel": rows[0].get("acq_kernel"),
                  "aggregate_bo_iterations_per_s": sum(r["aggregate_bo_iterations_per_s"] for r in rows),
                  "per_process": [round(r["aggregate_bo_iterations_per_s"]) for r in rows], "seconds": [round(r["seconds"], 1) for r in rows]}))
PY
exit $rc
