set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_t7.log 2>&1
echo "exit $?" >> gpurun_out/r3_t7.log
tail -12 gpurun_out/r3_t7.log
