#!/bin/bash
# A/B of compiler flags on the bench's kernels: builds the library with NXC_EXTRA_FLAGS="$1" and
# prints ms/pass of the fused kernel and of k_var at 1e6 / 1e7 packets; rebuild the product library
# afterwards (python3 -m nexoclom_amd.build --force).   bash tools/gpu_exp_flags.sh "<flags>"
NXC_EXTRA_FLAGS="$1" python3 -m nexoclom_amd.build --force > /dev/null || exit 1
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-h2d-pass 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
print('flags [$1]: fused %.2f ms, k_var 1e6 %.2f ms, 1e7 %.1f ms, tiles %.3f ms' % (d['roofline']['kernel_ms'], d['variable_step']['kernel_ms'], d['variable_step_1e7']['kernel_ms'], d['stored_samples_image']['tiles']['kernel_ms']))"
