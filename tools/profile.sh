#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then PMC passes (each in its own run, no
# trace domains beside --pmc).  Usage (on the box, from the repo root):
#   bash tools/profile.sh <tag> [extra bench args]
# Results land in gpurun_out/prof_<tag>/; tools/summarize_profile.py turns them into profiles/.
set -o pipefail
TAG=${1:-r01}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --no-h2d-pass: the pipelined upload pass needs two kernels side by side; counter mode serialises them
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-h2d-pass --steps 2 --warmup 1 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
grep '"metric"' $OUT/trace.log > $OUT/bench_line.json
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $BENCH > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
ls -R $OUT | head -60
