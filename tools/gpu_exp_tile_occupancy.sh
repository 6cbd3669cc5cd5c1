#!/bin/bash
# Pass 1 of the tiled image (k_image_bin) synchronises its whole workgroup twice per trip: one
# 1024-thread workgroup per CU (80 KB of staging) against two of 512 threads with half the staging
# each (chunks of 128 entries at 512 x 512).  tools/bench_kernels.py's tiled-image lines.
for P in "" "-DNXC_TILE_BIN_BLOCK_N=512 -DNXC_TILE_STAGE_N=4096" "-DNXC_TILE_BIN_BLOCK_N=512" "-DNXC_TILE_BIN_BLOCK_N=256 -DNXC_TILE_STAGE_N=2048"; do
  NXC_EXTRA_FLAGS="$P" python3 -m nexoclom_amd.build --force > /dev/null || exit 1
  echo "== flags $P"
  python3 tools/bench_kernels.py 2>&1 | grep "k_image_bin" | cut -c1-130
done
python3 -m nexoclom_amd.build --force > /dev/null
