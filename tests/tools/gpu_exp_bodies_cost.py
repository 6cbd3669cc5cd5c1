"""Cost of the moons / torus terms in the persistent kernel (1e6 Mercury-like packets)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from nexoclom_amd import hip_api
from oracle import np_oracle as O
from tests import helpers as H
from tests.test_gpu_bodies import bodies_cfg

f = H.mercury_forces('Na', 1.3)
n, endtime, step = 1_000_000, 50000., 30.
X0 = H.sample_x0(n, 1, endtime)
nsteps, n_iter = O.n_output_steps(endtime, step)
ctx = hip_api.Context(0)
H.set_ctx_forces(ctx, f)
a = np.array([3.0, 5.0])
moons = dict(gm=tuple(f.GM*np.array([1e-3, 1e-3])), radius=(0.01, 0.01), a=tuple(a),
             omega=tuple(np.sqrt(-f.GM/a**3)), phi=(0.3, 2.0))
cases = {
    'none': None,
    '2 moons': O.Bodies(t0=endtime, **moons),
    '1 moon': O.Bodies(t0=endtime, **{k: v[:1] for k, v in moons.items()}),
    'torus': O.Bodies(t0=endtime, chx_on=True, chx_k0=1e-6, chx_rho0=3., chx_width=1., chx_height=1.),
    'torus+vel': O.Bodies(t0=endtime, chx_on=True, chx_k0=1e-6, chx_rho0=3., chx_width=1., chx_height=1., chx_omega=1e-4),
    '2 moons+torus+vel': O.Bodies(t0=endtime, chx_on=True, chx_k0=1e-6, chx_rho0=3., chx_width=1., chx_height=1., chx_omega=1e-4, **moons),
}
for name, b in cases.items():
    ctx.set_bodies(bodies_cfg(b) if b is not None else None)
    ctx.upload_packets(X0)
    ms = []
    for _ in range(3):
        ctx.integrate_const_async(step, n_iter, 25., image=False)
        ctx.synchronize()
        ms.append(ctx.last_kernel_ms())
    w = ctx.counters()['particle_steps']
    print(f'{name:20s} {np.mean(ms[1:]):8.2f} ms  {w/np.mean(ms[1:])/1e6:7.2f} G p.s/s  ({w:.3e} steps)', flush=True)
ctx.set_bodies(None)
