#!/bin/bash
# The fair form of k_var with two (512-thread workgroups, one per CU) and four (1024) waves per SIMD
# instead of three.
for P in "-DNXC_BLOCK_PERSIST_N=512 -DNXC_VAR_ONE_WG_PER_CU" "-DNXC_BLOCK_PERSIST_N=1024"; do
NXC_EXTRA_FLAGS="$P" python3 -m nexoclom_amd.build --force > /dev/null || exit 1
NXC_TEST_VAR_VARIANT=fair python3 - <<PY
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench
from nexoclom_amd import Input, hip_api
ctx = hip_api.Context(0)
inputs = Input(bench.INFILE); inputs.options.step_size = 0.; inputs.options.resolution = 1e-4
for n in (250_000, 500_000, 1_000_000, 2_000_000):
    leg = bench.variable_leg(ctx, inputs, n, passes=3)
    print('flags "$P" fair: %9d packets  k_var %8.2f ms' % (n, leg['kernel_ms']), flush=True)
PY
done
python3 -m nexoclom_amd.build --force > /dev/null
