"""The reference's two-stage flow at the size of BASELINE configs[2]: Input.run(1e7) -- 125 Outputs of
80 467 packets in a few launch groups, 1.3e9 rows (53 GB) resident in HBM -- then produce_image over
the catalogue, against the streaming ModelImage of the same seeded packets."""
import contextlib, io, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, ModelImage, hip_api
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
ctx = hip_api.Context(0)
inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
free0 = ctx.mem_info()[0]
t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    inputs.run(n, seed=7, context=ctx)
t1 = time.time()
outs = inputs._catalogue
stores = {id(o._store): o._store for o in outs if o._store is not None}
rows = sum(o._nrows for o in outs)
params = {'quantity': 'radiance', 'dims': '512,512'}
with contextlib.redirect_stdout(io.StringIO()):
    two_stage = inputs.produce_image(params, context=ctx)
t2 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    streaming = ModelImage(inputs, params, npackets=len(outs)*len(outs[0]), packs_per_it=len(outs[0]),
                           seed=7, context=ctx)
t3 = time.time()
print(json.dumps({
    'Input.run': n, 'outputs': len(outs), 'launch_groups': len(stores), 'rows': rows,
    'rows_GB_in_HBM': sum(s.nbytes for s in stores.values())/1e9,
    'resident_outputs': sum(o.resident_rows(ctx) is not None for o in outs),
    'hbm_used_GB': (free0 - ctx.mem_info()[0])/1e9,
    'run_s': t1 - t0, 'produce_image_s': t2 - t1, 'streaming_ModelImage_s': t3 - t2,
    'packet_images_equal': bool(np.array_equal(two_stage.packet_image, streaming.packet_image)),
    'binned': float(two_stage.packet_image.sum()),
    'image_max_rel_diff': float(np.max(np.abs(two_stage.image - streaming.image))/streaming.image.max())}))

# again in the same process (the handle keeps the freed stores' blocks: no tens of GB of fresh
# device allocations through the driver the host's GPUs share), host-sampled
for o in inputs._catalogue:
    if o._store is not None:
        o._store.free()
inputs._catalogue.clear()
t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    inputs.run(n, seed=7, context=ctx)
t1 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    again = inputs.produce_image(params, context=ctx)
t2 = time.time()
print(json.dumps({'Input.run': n, 'sampler': 'numpy (host), second run of the process',
                  'launches': len({id(o._store) for o in inputs._catalogue}),
                  'run_s': t1 - t0, 'produce_image_s': t2 - t1,
                  'binned': float(again.packet_image.sum())}))
for o in inputs._catalogue:
    if o._store is not None:
        o._store.free()
del again

# the same packets drawn where they are integrated
del two_stage, streaming, outs, stores
inputs2 = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
inputs._catalogue.clear()
t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    inputs2.run(n, seed=7, context=ctx, sampler='device', generator='pcg64')
t1 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    img = inputs2.produce_image(params, context=ctx)
t2 = time.time()
print(json.dumps({'Input.run': n, 'sampler': "device, generator='pcg64'", 'run_s': t1 - t0,
                  'produce_image_s': t2 - t1, 'binned': float(img.packet_image.sum())}))
