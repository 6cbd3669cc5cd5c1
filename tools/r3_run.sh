set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_edge_fullsize.py -x -q -m gpu -k "input_run_rows" > gpurun_out/r3_t18.log 2>&1
echo "exit $?" >> gpurun_out/r3_t18.log
tail -25 gpurun_out/r3_t18.log
