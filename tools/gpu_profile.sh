#!/bin/bash
# Profiling recipe (run on the GPU box from the repo root; $1 = round tag, default r04): kernel statistics of the benchmark
# and of a 30-run batch with its DEFAULT worker threads, then three counter passes over fixed shapes (counters in their own
# runs, no tracing domains besides the kernel trace).  Copy what is to be judged from gpurun_out/<tag>/ into profiles/<tag>/.
set -e
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/bench_stats -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --batch 0 --no-kchol-grid > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
echo "bench stats done"
rocprofv3 --kernel-trace --stats -d $OUT/batch_stats -o batch30 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_batch_clock.py 30 40 > $OUT/batch30_under_rocprof.json 2> $OUT/batch30_under_rocprof.err
echo "batch stats done"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_mfma -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_pmc_shapes.py 2 > $OUT/pmc_mfma.log 2>&1
echo "pmc mfma done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_pmc_shapes.py 2 > $OUT/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_pmc_shapes.py 2 > $OUT/pmc_write.log 2>&1
echo "pmc write done"
cd $GRAFT_REPO_ROOT
python3 profiles/tools/summarise_pmc_passes.py $OUT > $OUT/pmc_summary.json
# keep the merge small: drop the per-dispatch traces (tens of MB), keep the statistics
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
ls -la $OUT $OUT/*/ | head -60
