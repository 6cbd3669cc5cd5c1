"""Debug helper: where do the HIP and C-oracle results differ for a bodies configuration?"""
import sys
import numpy as np
sys.path.insert(0, '.')
from nexoclom_amd import hip_api
from oracle import np_oracle as O
from oracle.c_oracle import COracle
from tests import helpers as H
from tests.test_gpu_bodies import random_bodies, bodies_cfg

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(500 + seed)
f = H.mercury_forces('Na', 1.3, True, bool(seed % 2 == 0), 0.0 if seed % 3 else 4000.0)
endtime, step = float(rng.choice([6000., 9000.])), float(rng.choice([30., 45.5]))
b = random_bodies(rng, f, endtime)
print(b)
X0 = H.sample_x0(1500, 10 + seed, endtime, vprob=3.0, delv=1.2)
nsteps, n_iter = O.n_output_steps(endtime, step)
n_iter = min(n_iter, nsteps - 1)
ctx = hip_api.Context(0)
co = COracle()
H.set_ctx_forces(ctx, f)
ctx.set_bodies(bodies_cfg(b))
ctx.upload_packets(X0)
t = ctx.integrate_const(step, n_iter, 9.0, nrec=nsteps, want_final=True)
ct = co.integrate_const(f, X0, step, n_iter, 9.0, nrec=nsteps, bodies=b)
d = t['traj'] != ct['traj']
print('differing entries', d.sum())
if d.any():
    c, k, i = np.argwhere(d)[np.argsort(np.argwhere(d)[:, 1])][0]
    print('first differing record: col', c, 'step', k, 'packet', i)
    print('prev state  ', ct['traj'][:, k-1, i])
    print('hip         ', t['traj'][:, k, i])
    print('oracle      ', ct['traj'][:, k, i])
    print('diff        ', t['traj'][:, k, i] - ct['traj'][:, k, i])
    for m in range(len(b.gm)):
        print('moon', m, [O.moon_xy(b, m, k-1, s, step) for s in (0, 5)])
