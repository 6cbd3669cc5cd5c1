#!/bin/bash
# PMC pass sq2 (VALU activity, LDS conflicts) over tools/gpu_experiments.py for a variant build of
# the library: bash tools/profile_variant.sh <tag> <lib.so>     (results: gpurun_out/prof_<tag>/)
TAG=$1; LIB=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export NEXOCLOM_HIP_LIB=$(realpath $LIB)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY --output-format csv -d $OUT/sq2 -- python3 $GRAFT_REPO_ROOT/tools/gpu_experiments.py 1e7 > $OUT/sq2.log 2>&1 || echo "pass failed"
grep "^n=" $OUT/sq2.log
