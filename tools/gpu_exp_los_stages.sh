#!/bin/bash
# What does each stage of k_los cost?  Builds the library with -DNXC_LOS_EXPERIMENT=N (results
# are WRONG in these builds: 1 = pairs inside the cones are counted but not weighed, 2 = no
# candidate queue / los_pair at all, 3 = group tests only, 4 = trip loads and spheres only, 5 = no trip at all: launch, tables, spectra) and
# times tools/bench_kernels.py's k_los line under rocprofv3.  On the GPU box, from the repo root:
#   bash tools/gpu_exp_los_stages.sh          (rebuild the product library afterwards)
set -o pipefail
for N in ${STAGES:-0 1 2 3 4 5}; do
  if [ $N = 0 ]; then NXC_EXTRA_FLAGS="" python3 -m nexoclom_amd.build --force > /dev/null
  else NXC_EXTRA_FLAGS="-DNXC_LOS_EXPERIMENT=$N" python3 -m nexoclom_amd.build --force > /dev/null; fi
  ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/los_stage$N -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py > $GRAFT_REPO_ROOT/gpurun_out/los_stage$N.log 2>&1 )
  python3 - <<PY
import csv,glob
f=sorted(glob.glob("$GRAFT_REPO_ROOT/gpurun_out/los_stage$N/**/*_kernel_stats.csv",recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if "k_los<float, int>" in r["Name"]: print("experiment $N: k_los<float,int> %.1f us" % (float(r["AverageNs"])/1e3))
PY
done
NXC_EXTRA_FLAGS="" python3 -m nexoclom_amd.build --force > /dev/null
