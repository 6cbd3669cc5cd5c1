set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_bodies.py -x -q -m gpu -k "shell or collinear" > gpurun_out/r3_t10.log 2>&1
echo "exit $?" >> gpurun_out/r3_t10.log
tail -40 gpurun_out/r3_t10.log
