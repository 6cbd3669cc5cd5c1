set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/r3_t17.log 2>&1
echo "exit $?" >> gpurun_out/r3_t17.log
tail -4 gpurun_out/r3_t17.log
grep -q "exit 0" gpurun_out/r3_t17.log && timeout -k 10 900 python tools/gpu_exp_variants.py nexoclom_amd/lib/libnexoclom_hip.so build/exp/libold7bf.so nexoclom_amd/lib/libnexoclom_hip.so build/exp/libold7bf.so nexoclom_amd/lib/libnexoclom_hip.so build/exp/libold7bf.so > gpurun_out/r3_variants6.log 2>&1
cat gpurun_out/r3_variants6.log
