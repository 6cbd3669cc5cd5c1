"""Ad-hoc timing experiments on the GPU box (not part of the test-suite)."""
import os, sys, time, io, contextlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, Output, ModelImage, hip_api
from nexoclom_amd.Output import n_output_steps

def setup(n, quantity='radiance', dims=512, seed=1234):
    inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
    ctx = hip_api.Context(0)
    with contextlib.redirect_stdout(io.StringIO()):
        out = Output(inputs, n, seed=seed, integrate=False, save=False, context=ctx)
        img = ModelImage(inputs, {'quantity': quantity, 'dims': f'{dims},{dims}'}, context=ctx)
    ctx.set_forces(**out.forces_kwargs())
    img._set_image(ctx, out.aplanet, out.vrplanet, True)
    return inputs, ctx, out, img

def timeit(ctx, fn, reps=3):
    fn(); ctx.synchronize()
    ms = []
    for _ in range(reps):
        fn(); ctx.synchronize(); ms.append(ctx.last_kernel_ms())
    return float(np.mean(ms))

if __name__ == '__main__':
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    inputs, ctx, out, img = setup(n)
    opt = inputs.options
    nsteps, n_iter = n_output_steps(opt.endtime.value, opt.step_size)
    soa = out.x0_soa()
    ctx.upload_soa(soa)
    t_img = timeit(ctx, lambda: (ctx.image_clear(), ctx.integrate_const_async(30., n_iter, 25., image=True)))
    work = ctx.counters()['particle_steps']
    t_no = timeit(ctx, lambda: ctx.integrate_const_async(30., n_iter, 25., image=False))
    print(f'n={n} work={work} fused+image {t_img:.2f} ms  ({work/t_img/1e6:.2f} Gps/s) | no image {t_no:.2f} ms ({work/t_no/1e6:.2f} Gps/s)')
    # sorted by speed descending (longest-lived first)
    v = np.sqrt(soa[4]**2 + soa[5]**2 + soa[6]**2)
    order = np.argsort(-v)
    ctx.upload_soa(np.ascontiguousarray(soa[:, order]))
    t_sorted = timeit(ctx, lambda: (ctx.image_clear(), ctx.integrate_const_async(30., n_iter, 25., image=True)))
    print(f'sorted by |v| desc: fused+image {t_sorted:.2f} ms ({work/t_sorted/1e6:.2f} Gps/s)')
    t_sorted_no = timeit(ctx, lambda: ctx.integrate_const_async(30., n_iter, 25., image=False))
    print(f'sorted by |v| desc: no image {t_sorted_no:.2f} ms ({work/t_sorted_no/1e6:.2f} Gps/s)')
