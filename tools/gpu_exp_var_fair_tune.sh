#!/bin/bash
# The fair form of k_var: length of a priority slice (2^SHIFT x 10 ns) and the most live lanes a
# donor wave may hand over.
for P in "-DNXC_VAR_PRIO_SHIFT=10" "-DNXC_VAR_PRIO_SHIFT=12" "-DNXC_VAR_PRIO_SHIFT=14" "-DNXC_VAR_PRIO_SHIFT=16" "-DNXC_VAR_MERGE_MAX_N=24" "-DNXC_VAR_MERGE_MAX_N=32"; do
  NXC_EXTRA_FLAGS="$P" python3 -m nexoclom_amd.build --force > /dev/null || exit 1
  NXC_TEST_VAR_VARIANT=fair python3 - <<PY
import contextlib, io, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench
from nexoclom_amd import Input, hip_api
ctx = hip_api.Context(0)
inputs = Input(bench.INFILE); inputs.options.step_size = 0.; inputs.options.resolution = 1e-4
out = []
for n in (500_000, 1_000_000, 2_000_000, 4_000_000):
    leg = bench.variable_leg(ctx, inputs, n, passes=3)
    out.append('%.2f' % leg['kernel_ms'])
print('flags "$P": k_var at 5e5 / 1e6 / 2e6 / 4e6 packets: ' + ' / '.join(out) + ' ms', flush=True)
PY
done
python3 -m nexoclom_amd.build --force > /dev/null
