set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r3_t21.log 2>&1
echo "exit $?" >> gpurun_out/r3_t21.log
tail -8 gpurun_out/r3_t21.log
