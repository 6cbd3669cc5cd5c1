set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r3_t16.log 2>&1
echo "exit $?" >> gpurun_out/r3_t16.log
tail -12 gpurun_out/r3_t16.log
