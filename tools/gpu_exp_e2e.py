"""End-to-end wall time of the public API (ModelImage streaming) for 1e7 packets."""
import os, sys, io, contextlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, ModelImage, hip_api
inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
ctx = hip_api.Context(0)
params = {'quantity': 'radiance', 'dims': '512,512'}
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
for sampler in ('device', 'device', 'numpy'):
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        img = ModelImage(inputs, params, npackets=n, seed=1, context=ctx, sampler=sampler)
    dt = time.time() - t0
    print(f'ModelImage(npackets={n}, sampler={sampler}): {dt:.3f} s wall, '
          f'{img.counters["particle_steps"]/dt/1e9:.2f} G particle*steps/s end to end, image sum {img.image.sum():.6e}')
