"""What the queue order is worth to the fused pass: speed order (what an upload can know) against
the exact-lifetime order a counting pass leaves behind (the upper bound of any predictor)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gpu_experiments import setup, timeit
from nexoclom_amd.Output import n_output_steps
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
inputs, ctx, out, img = setup(n)
opt = inputs.options
nsteps, n_iter = n_output_steps(opt.endtime.value, opt.step_size)
soa = out.x0_soa()
ctx.upload_soa(soa)
run = lambda: (ctx.image_clear(), ctx.integrate_const_async(30., n_iter, 25., image=True))
noimg = lambda: ctx.integrate_const_async(30., n_iter, 25., image=False)
print(f'speed order:    image {timeit(ctx, run):.2f} ms, no image {timeit(ctx, noimg):.2f} ms', flush=True)
lib = ctx.lib
import ctypes as C
lengths = np.empty(n, dtype=np.int64); total = C.c_int64(0)
ctx._check(lib.nxc_integrate_const_rows(ctx._h, C.c_double(30.), C.c_int64(n_iter), C.c_double(25.),
                                        lengths.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(total)))
print(f'lifetime order: image {timeit(ctx, run):.2f} ms, no image {timeit(ctx, noimg):.2f} ms '
      f'(rows {total.value:.3e})', flush=True)
ctx.upload_soa(soa)
print(f'speed order:    image {timeit(ctx, run):.2f} ms, no image {timeit(ctx, noimg):.2f} ms', flush=True)
