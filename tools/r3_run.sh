set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/r3_t12.log 2>&1
echo "exit $?" >> gpurun_out/r3_t12.log
tail -4 gpurun_out/r3_t12.log
grep -q "exit 0" gpurun_out/r3_t12.log && timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3_bench3.json 2> gpurun_out/r3_bench3.err
echo "exit $?"; python -c "
import json; l=json.load(open('gpurun_out/r3_bench3.json')); print(l['ms_per_step'], l['ms_per_step_incl_h2d'], l['h2d_pass'], l['roofline']['frac'])"
bash tools/profile.sh r03d > gpurun_out/r3_prof_d.log 2>&1
echo "profile d exit $?"
