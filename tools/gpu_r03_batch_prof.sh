#!/bin/bash
# Round 3: the run that aborted in round 2 (gpurun_out/bclock30_prof.log) ONCE more, with the default worker threads, under
# rocprofv3 --kernel-trace, with tools/segv_maps.so in front: either it aborts again and the shim says where (module + offset
# of every frame, read or write, the mappings around the address), or it completes and its kernel statistics replace
# profiles/r02/batch30_kernel_stats.csv (which described a one-worker execution).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03/batch_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LD_PRELOAD=$GRAFT_REPO_ROOT/tools/segv_maps.so timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $OUT -o batch30 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_batch_clock.py 30 40 > $OUT/batch30_under_rocprof.json 2> $OUT/batch30_under_rocprof.err
echo "exit $?" | tee $OUT/exit.txt
grep -n "segv_maps" $OUT/batch30_under_rocprof.err | head -80
find $OUT -name "*kernel_trace.csv" -size +2M -delete
ls -la $OUT
tail -3 $OUT/batch30_under_rocprof.json | cut -c1-600
