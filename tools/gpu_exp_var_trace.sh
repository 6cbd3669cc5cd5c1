#!/bin/bash
# per-wave timeline of k_var at 1e6 packets, in its two launch forms
NXC_EXTRA_FLAGS="-DNXC_VAR_TRACE" python3 -m nexoclom_amd.build --force > /dev/null || exit 1
for V in plain fair; do
  echo "== $V"
  NXC_TEST_VAR_VARIANT=$V python3 tools/gpu_exp_var_trace.py 1e6 || exit 1
done
python3 -m nexoclom_amd.build --force > /dev/null
