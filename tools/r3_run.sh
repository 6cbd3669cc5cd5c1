set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_bodies.py -x -q -m gpu > gpurun_out/r3_t4.log 2>&1
echo "exit $?" >> gpurun_out/r3_t4.log
tail -30 gpurun_out/r3_t4.log
timeout -k 10 300 python tools/gpu_exp_twostage.py > gpurun_out/r3_twostage1.log 2>&1
tail -45 gpurun_out/r3_twostage1.log | cut -c1-180
