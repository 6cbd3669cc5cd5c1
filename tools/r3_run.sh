set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/r3_bench2.json 2> gpurun_out/r3_bench2.err
echo "bench exit $?"; cut -c1-300 gpurun_out/r3_bench2.json; tail -2 gpurun_out/r3_bench2.err
timeout -k 10 300 python tools/gpu_exp_rows.py 13 > gpurun_out/r3_rows3.log 2>&1; cat gpurun_out/r3_rows3.log
bash tools/profile.sh r03b > gpurun_out/r3_prof_b.log 2>&1
echo "profile b exit $?"
bash tools/profile_kernels.sh r03l > gpurun_out/r3_prof_l.log 2>&1
echo "profile l exit $?"
