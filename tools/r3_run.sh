set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed or rccl or resident or compact" > gpurun_out/r3_t9.log 2>&1
echo "exit $?" >> gpurun_out/r3_t9.log
tail -30 gpurun_out/r3_t9.log
grep -q "exit 0" gpurun_out/r3_t9.log && timeout -k 10 400 python bench.py --with-comm --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3_bench_withcomm.json 2> gpurun_out/r3_bench_withcomm.err
echo "exit $?"; cut -c1-800 gpurun_out/r3_bench_withcomm.json
