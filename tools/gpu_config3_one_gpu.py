"""BASELINE configs[3] at full size on ONE GPU: 1e8 packets (Na @ Mercury, 1667 steps, fused 512^2
radiance image) as eight index shards of 1.25e7 integrated one after the other and summed on the
host -- the sum the RCCL reduce forms on eight GPUs (ModelImage.py:96-98 of the reference) -- against
the same 1e8 packets in one piece.  Packet-count images must be identical, the weight images equal
to fp64 summation order."""
import contextlib, io, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, ModelImage, hip_api
from nexoclom_amd.distributed import shard_range

total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
world = 8
ctx = hip_api.Context(0)
inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
params = {'quantity': 'radiance', 'dims': '512,512'}
for sampler, generator in (('device', 'pcg64'), ('device', 'philox'), ('numpy', None)):
    kw = dict(npackets=total, seed=1234, context=ctx, sampler=sampler)
    if generator:
        kw['generator'] = generator
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        whole = ModelImage(inputs, params, **kw)
    t1 = time.time()
    image = np.zeros_like(whole.image)
    counts = np.zeros_like(whole.packet_image)
    steps = 0
    source = 0.0
    shard_s = []
    for rank in range(world):
        lo, hi = shard_range(total, rank, world)
        t = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            part = ModelImage(inputs, params, shard=(lo, hi), finalize=False, **kw)
        shard_s.append(round(time.time() - t, 3))
        image += part.image
        counts += part.packet_image
        steps += part.counters['particle_steps']
        source += part.totalsource
    per_second = source/inputs.options.endtime.value
    image *= 1e23/per_second
    print(json.dumps({
        'config': 'BASELINE configs[3] on one GPU', 'packets': total, 'shards': world,
        'sampler': sampler, 'generator': generator,
        'one_piece_s': round(t1 - t0, 3), 'shard_s': shard_s,
        'particle_steps_one_piece': int(whole.counters['particle_steps']),
        'particle_steps_shards': int(steps),
        'packet_images_identical': bool(np.array_equal(counts, whole.packet_image)),
        'binned': float(counts.sum()),
        'image_max_rel_diff': float(np.max(np.abs(image - whole.image))/whole.image.max())}),
        flush=True)
