"""A/B of a library build on the bench pass (1e7 packets, fused 512^2 image, float32 down-cast as
ModelImage streams it): times, counters, and the image pair saved for comparison with another
build's.  python tools/gpu_exp_ab_image.py out.npz [quantity]  (library via NEXOCLOM_HIP_LIB)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gpu_experiments import setup, timeit
from nexoclom_amd.Output import n_output_steps
quantity = sys.argv[2] if len(sys.argv) > 2 else 'radiance'
inputs, ctx, out, img = setup(10_000_000, quantity=quantity)
opt = inputs.options
nsteps, n_iter = n_output_steps(opt.endtime.value, opt.step_size)
ctx.upload_soa(out.x0_soa())
t = [timeit(ctx, lambda: (ctx.image_clear(), ctx.integrate_const_async(30., n_iter, 25., image=True)))
     for _ in range(3)]
ctx.image_clear()
ctx.integrate_const_async(30., n_iter, 25., image=True)
ctx.synchronize()
image, counts = ctx.image_download()
c = ctx.counters()
np.savez(sys.argv[1], image=image, counts=counts, ctr=np.array([c[k] for k in sorted(c)]))
print(os.environ.get('NEXOCLOM_HIP_LIB', 'product'), quantity, ' '.join(f'{x:.2f}' for x in t), 'ms', c)
