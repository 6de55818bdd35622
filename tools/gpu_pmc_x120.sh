#!/bin/bash
# Counter passes (MFMA, FETCH_SIZE, WRITE_SIZE - each in its own run, kernel trace only) over the conditioning of 120 runs at
# (450, 36) and (1050, 89): what the oversubscribed look-back / panel / root-inverse kernels pull through the fabric.
# usage (GPU box, repo root): tools/gpu_pmc_x120.sh <round tag>
set -e
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/x120
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_mfma -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_pmc_shapes.py 2 120 > $OUT/pmc_mfma.log 2>&1
echo "pmc mfma done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_pmc_shapes.py 2 120 > $OUT/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_pmc_shapes.py 2 120 > $OUT/pmc_write.log 2>&1
echo "pmc write done"
cd $GRAFT_REPO_ROOT
python3 profiles/tools/summarise_pmc_passes.py $OUT > $OUT/pmc_summary_x120.json
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
