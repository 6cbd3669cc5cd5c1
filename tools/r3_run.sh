set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python tools/gpu_exp_run1e7.py 1e7 > gpurun_out/r3_run1e7.log 2>&1
tail -5 gpurun_out/r3_run1e7.log | cut -c1-900
