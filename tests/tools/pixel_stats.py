"""Pixel statistics of the bench workload's image samples (the numbers DESIGN.md section 3 quotes
when it argues for global pair-atomics over LDS-privatised tiles): computed from the packet-count
image of the C oracle (TEST INFRASTRUCTURE, hence under tests/tools/) on 1e6 packets of BASELINE
configs[2]; runs on the CPU.   python tests/tools/pixel_stats.py > profiles/r02_pixel_stats.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np_oracle as O          # noqa: E402
from oracle.c_oracle import COracle        # noqa: E402
from tests import helpers as H             # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
f = H.mercury_forces('Na', 1.3)
X0 = H.sample_x0(n, 1234, 50000.)
nsteps, n_iter = O.n_output_steps(50000., 30.)
co = COracle()
im = H.image_setup(f, 'radiance', dims=(512, 512))
desc = co.image_desc(im['M'], f.vrplanet, im['apix'], 'radiance', im['g_tables'], im['xedges'],
                     im['zedges'], downcast=True)
res = co.integrate_const(f, X0, 30., n_iter, 25., img=desc, threads=co.max_threads())
counts = res['counts'].astype(np.float64)
binned = counts.sum()
samples = float(res['steps'].sum() + n)        # records offered: every live record incl. record 0
p = counts.ravel()/binned
lines = counts.reshape(512, 128, 4).sum(axis=2).ravel()/binned      # 64-byte lines: 4 pixels of 16 B
top = np.sort(counts.ravel())[::-1]
out = {
    'workload': f'BASELINE configs[2] forces and source, {n} packets, 512x512 image, width 8x8 R',
    'particle_steps': int(res['work']),
    'samples_inside_image': int(binned),
    'fraction_of_samples_inside_image': binned/samples,
    'sum_p2_over_pixels': float((p**2).sum()),
    'sum_p2_over_64B_lines': float((lines**2).sum()),
    'expected_same_pixel_pairs_in_64_lane_instruction': float(64*63/2*(p**2).sum()),
    'expected_same_line_pairs_in_64_lane_instruction': float(64*63/2*(lines**2).sum()),
    'share_of_samples_in_hottest_12000_pixels': float(top[:12000].sum()/binned),
    'share_of_samples_in_hottest_4096_pixels': float(top[:4096].sum()/binned),
    'pixels_touched': int((counts > 0).sum()),
    'mean_steps_per_packet': float(res['steps'].mean()),
    'median_steps_per_packet': float(np.median(res['steps'])),
}
print(json.dumps(out, indent=1))
