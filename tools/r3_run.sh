set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_variant.sh r03lut_product nexoclom_amd/lib/libnexoclom_hip.so
bash tools/profile_variant.sh r03lut_3stage build/exp/liblut3.so
