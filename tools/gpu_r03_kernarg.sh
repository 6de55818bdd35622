#!/bin/bash
# Round 3: the stand-alone model of the abort in gpurun_out/bclock30_prof.log (profiles/tools/kernarg_threads.hip), plain
# and under rocprofv3 --kernel-trace, one argument size per process.  Stops at the first step that had to be killed.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03/kernarg
mkdir -p $OUT
BIN=$GRAFT_REPO_ROOT/profiles/tools/kernarg_threads
cd /tmp && export TMPDIR=/tmp
for N in 64 256 400 448; do
  timeout -k 10 120 $BIN 8 20000 $N > $OUT/plain_$N.log 2>&1; rc=$?
  echo "plain $N: exit $rc" | tee -a $OUT/summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
for N in 64 256 400 448; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$N -o k --output-format csv -- $BIN 8 20000 $N > $OUT/prof_$N.log 2>&1; rc=$?
  echo "rocprofv3 --kernel-trace $N: exit $rc" | tee -a $OUT/summary.txt
  tail -3 $OUT/prof_$N.log >> $OUT/summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  find $OUT/prof_$N -name "*kernel_trace.csv" -delete
done
# one worker thread under the profiler: is it the threads or the size?
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_448_t1 -o k --output-format csv -- $BIN 1 160000 448 > $OUT/prof_448_t1.log 2>&1; rc=$?
echo "rocprofv3 --kernel-trace 448, one thread: exit $rc" | tee -a $OUT/summary.txt
tail -3 $OUT/prof_448_t1.log >> $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/summary.txt
