"""Runs in a process of its own with NEXOCLOM_HIP_LIB pointing at libnexoclom_hip_2r.so (the kernels
built with -DNXC_TABLEAU_TWO_ROUNDINGS: NumPy's two roundings per tableau term) and compares them with
the C checker built the same way -- bit for bit, like the product pair -- and with the NumPy oracle
(the reference's arithmetic).  Started by tests/test_gpu_two_roundings.py; prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nexoclom_amd import hip_api                      # noqa: E402
from oracle import np_oracle as O                     # noqa: E402
from oracle.c_oracle import COracle                   # noqa: E402
from tests import helpers as H                        # noqa: E402

assert hip_api.LIB_PATH.endswith('libnexoclom_hip_2r.so'), hip_api.LIB_PATH
ctx = hip_api.Context(0)
two, fused = COracle(two_roundings=True), COracle()
f = H.mercury_forces('Na', 1.3)
H.set_ctx_forces(ctx, f)
out = {}
# one step
X = H.random_cloud(4096, 9)
h = np.random.default_rng(3).uniform(1, 120, len(X))
g, gd = ctx.rk5_step(X, h, want_delta=True)
c, cd = two.rk5(f, X, h, want_delta=True)
ref, _ = O.rk5(f, X, h, want_delta=True)
out['step_bit_exact'] = bool(np.array_equal(g, c) and np.array_equal(gd, cd))
out['step_differs_from_fused'] = bool(not np.array_equal(g, fused.rk5(f, X, h)[0]))
out['step_vs_numpy'] = float((np.abs(g - ref)/np.maximum(np.abs(ref), 1e-3)).max())
# the constant driver with the image, and the adaptive driver
n, endtime, step, edge = 20000, 50000., 30., 25.
X0 = H.sample_x0(n, 4711, endtime)
nsteps, n_iter = O.n_output_steps(endtime, step)
im = H.image_setup(f, 'radiance', dims=(128, 128))
ctx.set_image(im['M'], f.vrplanet, im['apix'], 'radiance', im['xedges'], im['zedges'], im['g_tables'],
              downcast_f32=True)
ctx.upload_packets(X0)
res = ctx.integrate_const(step, n_iter, edge, image=True, want_final=True, want_steps=True)
image, counts = ctx.image_download()
desc = two.image_desc(im['M'], f.vrplanet, im['apix'], 'radiance', im['g_tables'], im['xedges'],
                      im['zedges'], downcast=True)
cc = two.integrate_const(f, X0, step, n_iter, edge, img=desc, threads=two.max_threads())
out['const_bit_exact'] = bool(np.array_equal(res['final'], cc['final']) and
                              np.array_equal(res['steps'], cc['steps']) and
                              np.array_equal(counts, cc['counts']))
out['const_work'] = int(ctx.counters()['particle_steps']) == cc['work']
out['image_rel'] = float(np.max(np.abs(image - cc['image'])[cc['image'] > 0]/cc['image'][cc['image'] > 0]))
Xv = H.sample_x0(3000, 616, 20000.)
Xv[:, 0] = np.random.default_rng(61).random(3000)*20000.
ctx.upload_packets(Xv)
fin, hs = ctx.integrate_var(1e-4, 25.0)
cfin, chs, cwork, bad = two.integrate_var(f, Xv, 1e-4, 25.0)
out['var_bit_exact'] = bool(np.array_equal(fin, cfin) and np.array_equal(hs, chs))
out['var_work'] = int(ctx.counters()['particle_steps']) == cwork
print(json.dumps(out), flush=True)
ctx.close()
