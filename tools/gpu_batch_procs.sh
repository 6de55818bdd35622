#!/bin/bash
# P processes x (30 / P) runs each on ONE GPU (batched lock-step inside each process): aggregate rate.  usage: gpu_batch_procs.sh P [threads]
P=${1:-2}; T=${2:-4}
R=$((30 / P))
rm -f gpurun_out/bp_*.log
for i in $(seq 1 $P); do
  PCABO_BATCH_THREADS=$T python tools/gpu_batch_clock.py $R 40 2>/dev/null | tail -1 > gpurun_out/bp_$i.log &
done
wait
rm -f gpurun_out/bp_sum.txt
cat gpurun_out/bp_*.log | python -c "
import sys, json
tot=0
for l in sys.stdin:
    d=json.loads(l)
    tot+=d['aggregate_bo_iterations_per_s']; print(d['runs'], d['aggregate_bo_iterations_per_s'], d['seconds'])
print('SUM', tot)
"
