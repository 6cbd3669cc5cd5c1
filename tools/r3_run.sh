set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/r3_t1.log 2>&1
echo "exit $?" >> gpurun_out/r3_t1.log
tail -15 gpurun_out/r3_t1.log
