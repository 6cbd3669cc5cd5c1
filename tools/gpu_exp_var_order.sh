#!/bin/bash
NXC_EXTRA_FLAGS="-DNXC_VAR_TRACE" python3 -m nexoclom_amd.build --force > /dev/null || exit 1
python3 tools/gpu_exp_var_order.py 1e6
python3 -m nexoclom_amd.build --force > /dev/null
