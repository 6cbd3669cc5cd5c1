set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "batched" > gpurun_out/r3_t22.log 2>&1
echo "exit $?" >> gpurun_out/r3_t22.log
tail -25 gpurun_out/r3_t22.log
