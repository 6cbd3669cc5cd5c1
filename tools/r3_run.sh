set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python tools/gpu_exp_variants.py nexoclom_amd/lib/libnexoclom_hip.so build/exp/liblut1.so build/exp/libold7bf.so nexoclom_amd/lib/libnexoclom_hip.so build/exp/liblut1.so build/exp/libold7bf.so nexoclom_amd/lib/libnexoclom_hip.so build/exp/liblut1.so build/exp/libold7bf.so > gpurun_out/r3_variants5.log 2>&1
cat gpurun_out/r3_variants5.log
