set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r3_bench_2rank.out 2> gpurun_out/r3_bench_2rank.err
echo "exit code $?" | tee -a gpurun_out/r3_bench_2rank.out
cat gpurun_out/r3_bench_2rank.out | cut -c1-600
grep "bench rank" gpurun_out/r3_bench_2rank.err | head
