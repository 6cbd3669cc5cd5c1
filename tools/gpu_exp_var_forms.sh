#!/bin/bash
# The two launch forms of k_var (NXC_TEST_VAR_VARIANT: fair = clock-rotated wave priorities and a
# merged tail, plain) from 2.5e5 to 1e7 packets: where nxc_integrate_var's threshold (24 packets per
# lane) comes from.
for V in fair plain; do
NXC_TEST_VAR_VARIANT=$V python3 - <<PY
import contextlib, io, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench
from nexoclom_amd import Input, hip_api
ctx = hip_api.Context(0)
inputs = Input(bench.INFILE); inputs.options.step_size = 0.; inputs.options.resolution = 1e-4
for n in (250_000, 500_000, 1_000_000, 2_000_000, 4_000_000, 6_000_000, 10_000_000):
    leg = bench.variable_leg(ctx, inputs, n, passes=2)
    print('$V: %9d packets (%5.1f per lane)  k_var %8.2f ms  %.3g attempts/s' % (n, n/(256*768), leg['kernel_ms'], leg['value']), flush=True)
PY
done
