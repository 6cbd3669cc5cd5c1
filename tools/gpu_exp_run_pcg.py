"""Input.run(n) with the device following the host sampler's PCG64 streams, in a fresh process
(twice: the second run reuses the handle's pooled blocks)."""
import contextlib, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, hip_api
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
ctx = hip_api.Context(0)
for rep in range(2):
    inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        inputs.run(n, seed=7, context=ctx, sampler='device', generator='pcg64')
    t1 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        img = inputs.produce_image({'quantity': 'radiance', 'dims': '512,512'}, context=ctx)
    print(json.dumps({'Input.run': n, 'generator': 'pcg64', 'rep': rep, 'run_s': t1 - t0,
                      'produce_image_s': time.time() - t1, 'binned': float(img.packet_image.sum())}), flush=True)
    for o in inputs._catalogue:
        if o._store is not None:
            o._store.free()
