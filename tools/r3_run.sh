set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_t19.log 2>&1
echo "exit $?" >> gpurun_out/r3_t19.log
tail -5 gpurun_out/r3_t19.log
timeout -k 10 600 python bench.py > gpurun_out/r3_bench4.json 2> gpurun_out/r3_bench4.err
echo "bench exit $?"
bash tools/profile.sh r03e > gpurun_out/r3_prof_e.log 2>&1
echo "profile e exit $?"
bash tools/profile_kernels.sh r03m > gpurun_out/r3_prof_m.log 2>&1
echo "profile m exit $?"
