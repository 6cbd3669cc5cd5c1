set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tools/gpu_exp_los_e2e.py 1e6 512 > gpurun_out/r3_los_e2e.log 2>&1
head -12 gpurun_out/r3_los_e2e.log | cut -c1-160
