"""SURVEY 8(d)'s stress vector (no early deaths) through the fused kernel, with and without the image."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gpu_experiments import setup, timeit
from nexoclom_amd.Output import n_output_steps
inputs, ctx, out, img = setup(1000)
opt = inputs.options
nsteps, n_iter = n_output_steps(opt.endtime.value, opt.step_size)
ns = 5*256*768
rng = np.random.default_rng(5)
phi = rng.uniform(0, 2*np.pi, ns)
stress = np.zeros((8, ns))
stress[0] = opt.endtime.value
stress[1], stress[2], stress[3] = 30*np.cos(phi), rng.uniform(-5, 5, ns), 30*np.sin(phi)
stress[5] = rng.uniform(-2, 2, ns)/out.unit_km
stress[7] = 1.0
ctx.upload_soa(stress)
t_no = timeit(ctx, lambda: ctx.integrate_const_async(30., n_iter, 1e9, image=False))
t_im = timeit(ctx, lambda: (ctx.image_clear(), ctx.integrate_const_async(30., n_iter, 1e9, image=True)))
print(f'stress: no image {t_no:.2f} ms, image {t_im:.2f} ms')
