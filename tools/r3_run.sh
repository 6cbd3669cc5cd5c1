set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in nexoclom_amd/lib/libnexoclom_hip.so build/exp/liblos4.so build/exp/liblos16.so build/exp/liblos32.so; do
  echo $lib; NEXOCLOM_HIP_LIB=$lib timeout -k 10 300 python tools/bench_kernels.py 2>&1 | grep '"k_los"' | cut -c1-120
done
