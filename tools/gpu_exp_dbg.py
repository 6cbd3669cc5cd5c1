import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys; sys.path.insert(0, %r)
from tools.gpu_experiments import *
n=int(float(sys.argv[1]))
inputs, ctx, out, img = setup(n)
opt=inputs.options; nsteps,n_iter=n_output_steps(opt.endtime.value,opt.step_size)
ctx.upload_soa(out.x0_soa())
t=timeit(ctx, lambda:(ctx.image_clear(), ctx.integrate_const_async(30.,n_iter,25.,image=True)))
print("dbg", os.environ.get("NXC_DEBUG_IMAGE","0"), "fused+image %%.2f ms" %% t, ctx.counters())
''' % ROOT
for dbg in ('0', '1', '2', '3'):
    env = dict(os.environ, NXC_DEBUG_IMAGE=dbg)
    r = subprocess.run([sys.executable, '-c', code, sys.argv[1] if len(sys.argv) > 1 else '1e7'], env=env, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
