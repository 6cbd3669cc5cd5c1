"""The pipelined upload pass by number of pieces: wall time and kernel time (HIP events around the
persistent launch, which includes its waiting for the first piece)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gpu_experiments import setup
from nexoclom_amd.Output import n_output_steps
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
inputs, ctx, out, img = setup(n)
opt = inputs.options
nsteps, n_iter = n_output_steps(opt.endtime.value, opt.step_size)
soa = out.x0_soa()
ctx.upload_soa(soa)
for rep in range(2):
    ctx.image_clear(); ctx.integrate_const_async(30., n_iter, 25., image=True); ctx.synchronize()
print(f'resident pass: {ctx.last_kernel_ms():.2f} ms', flush=True)
for pieces in (1, 2, 4, 8, 16, 32):
    for rep in range(2):
        ctx.image_clear(); ctx.synchronize()
        t0 = time.perf_counter()
        ctx.integrate_const_streamed(soa, 30., n_iter, 25., image=True, pieces=pieces)
        ctx.synchronize()
        wall = (time.perf_counter() - t0)*1e3
    c = ctx.counters()
    print(f'pieces {pieces:2d}: wall {wall:7.2f} ms, kernel {ctx.last_kernel_ms():7.2f} ms, '
          f'unfinished {c["unfinished"]}, steps {c["particle_steps"]}', flush=True)
