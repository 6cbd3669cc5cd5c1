set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r3_t15.log 2>&1
echo "exit $?" >> gpurun_out/r3_t15.log
tail -5 gpurun_out/r3_t15.log
timeout -k 10 300 python tools/gpu_exp_twostage.py > gpurun_out/r3_twostage4.log 2>&1
grep "Input.run" gpurun_out/r3_twostage4.log | cut -c1-200
