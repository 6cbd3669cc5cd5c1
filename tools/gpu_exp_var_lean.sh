#!/bin/bash
# k_var compiled for 768 threads (three waves per SIMD) against k_var compiled for 256 (one wave per
# SIMD, better interleaved chains) by number of packets: where does the lean variant stop paying?
# Builds twice (NXC_VAR_LEAN_PACKETS_PER_LANE_N = 0: never lean, 100000: always); rebuild afterwards.
for L in 0 100000; do
  NXC_EXTRA_FLAGS="-DNXC_VAR_LEAN_PACKETS_PER_LANE_N=$L" python3 -m nexoclom_amd.build --force > /dev/null || exit 1
  python3 - <<PY
import contextlib, io, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench
from nexoclom_amd import Input, hip_api
ctx = hip_api.Context(0)
inputs = Input(bench.INFILE); inputs.options.step_size = 0.; inputs.options.resolution = 1e-4
for n in (250_000, 500_000, 1_000_000, 2_000_000, 4_000_000, 10_000_000):
    leg = bench.variable_leg(ctx, inputs, n, passes=2)
    print('lean threshold $L: %9d packets  k_var %8.2f ms  %.3g attempts/s' % (n, leg['kernel_ms'], leg['value']), flush=True)
PY
done
python3 -m nexoclom_amd.build --force > /dev/null
