"""k_image (one global atomic pair per binned sample) against the tiled image (k_image_bin +
k_image_tiles) over the resident rows of an Input.run: kernel time by HIP events, counters, and the
two images compared.  python tools/gpu_exp_image_tiles.py [npackets] [quantity]"""
import contextlib, io, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nexoclom_amd
from nexoclom_amd import Input, hip_api

n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
quantity = sys.argv[2] if len(sys.argv) > 2 else 'radiance'
infile = os.path.join(os.path.dirname(nexoclom_amd.__file__), 'inputfiles', 'Na.mercury.bench.input')
ctx = hip_api.Context(0)
inputs = Input(infile)
with contextlib.redirect_stdout(io.StringIO()):
    inputs.run(n, seed=7, context=ctx, sampler='device', generator='pcg64')
rows = sum(o._nrows for o in inputs._catalogue)
print(f'Input.run({n:g}): {len(inputs._catalogue)} Outputs, {rows:.4e} rows resident', flush=True)
params = {'quantity': quantity, 'dims': '512,512'}
out = {}
for mode, args in (('atomics', ()), ('tiles', ()), ('tiles', (0, 1 << 26)), ('tiles', (0, 1 << 28)),
                   ('atomics', ()), ('tiles', ())):
    ctx.image_mode(mode, *args)
    best = None
    for rep in range(3):
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            img = inputs.produce_image(params, context=ctx)
        wall = time.time() - t0
        best = wall if best is None else min(best, wall)
    ms = getattr(img, 'kernel_ms', None)
    key = mode + (f'/slab=2^{int(np.log2(args[1]))}' if args and args[1] else '')
    out.setdefault(key, dict(wall_s=best, image=img.image.copy(), counts=img.packet_image.copy()))
    print(f'{key:18s} produce_image wall {best*1e3:8.2f} ms   last kernel {ctx.last_kernel_ms():7.3f} ms',
          flush=True)
ctx.image_mode('auto')
a, t = out['atomics'], out['tiles']
same = bool(np.array_equal(a['counts'], t['counts']))
rel = float(np.max(np.abs(a['image'] - t['image'])/np.maximum(np.abs(a['image']), 1e-300)))
print(json.dumps({'rows': rows, 'quantity': quantity, 'atomics_wall_ms': a['wall_s']*1e3,
                  'tiles_wall_ms': t['wall_s']*1e3, 'packet_counts_identical': same,
                  'binned': float(t['counts'].sum()), 'max_rel_diff_image': rel}))
