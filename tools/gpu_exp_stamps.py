"""Where a trip through the persistent loop spends its cycles: s_memtime shares per segment from a
diagnostic build (hipcc ... -DNXC_STAMPS -o variants/lib_stamps.so).  The stamps forbid
overlaps the product kernel has, so read the SHARES, not the run time.
    NEXOCLOM_HIP_LIB=variants/lib_stamps.so python tools/gpu_exp_stamps.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gpu_experiments import setup, timeit
from nexoclom_amd.Output import n_output_steps
inputs, ctx, out, img = setup(int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000)
opt = inputs.options
nsteps, n_iter = n_output_steps(opt.endtime.value, opt.step_size)
ctx.upload_soa(out.x0_soa())
names = ['refill', 'step+fate', 'locate', 'bookkeeping', 'push', 'pop+weight', 'atomics', '-']
for image in (True, False):
    ctx.image_clear(); ctx.integrate_const_async(30., n_iter, 25., image=image); ctx.synchronize()
    ms = ctx.last_kernel_ms()
    buf = (C.c_ulonglong*8)()
    ctx.lib.nxc_debug_stamps(ctx._h, buf)
    tot = sum(buf)
    print(f'image={image}: kernel {ms:.2f} ms (stamped build)')
    for n, v in zip(names, buf):
        print(f'   {n:12s} {v:16d} cycles  {100*v/max(tot,1):5.1f} %')
