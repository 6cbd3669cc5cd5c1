"""Large resident sets through the public API: 1e8 device-sampled packets in one ModelImage call
(five 2e7-packet chunks), the reference-run sources at 2e7, and one 1.25e7-packet shard of
BASELINE configs[3] through sharded_image's own code path (world of one)."""
import os, sys, io, contextlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, ModelImage, hip_api
from nexoclom_amd.distributed import ControlPlane, sharded_image, shard_range
ctx = hip_api.Context(0)
params = {'quantity': 'radiance', 'dims': '512,512'}
def run(label, fn):
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        img = fn()
    dt = time.time() - t0
    print(f'{label}: {dt:.2f} s wall, {img.npackets} packets, {img.counters["particle_steps"]:.3e} '
          f'particle*steps ({img.counters["particle_steps"]/dt/1e9:.1f} G/s end to end), '
          f'{int(img.packet_image.sum())} samples binned', flush=True)
bench = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
run('bench source, 1e8 device-sampled', lambda: ModelImage(bench, params, npackets=100_000_000, seed=1, context=ctx, sampler='device'))
ref = Input(os.path.join(ROOT, 'tests', 'golden', 'inputfiles', 'Na.reference.input'))
run('Na.reference.input (surface spot + maxwellian), 2e7 device-sampled', lambda: ModelImage(ref, params, npackets=20_000_000, seed=1, context=ctx, sampler='device'))
lo, hi = shard_range(100_000_000, 3, 8)
run(f'configs[3] shard 3 of 8 ([{lo}, {hi}) of 1e8), device sampler', lambda: ModelImage(bench, params, npackets=100_000_000, shard=(lo, hi), seed=1, context=ctx, sampler='device', finalize=False))
run('sharded_image world=1, 2e7 host-sampled (chunk grid of 1e7)', lambda: sharded_image(bench, params, 20_000_000, 7, cp=ControlPlane(1, 0), context=ctx, sampler='numpy', packs_per_it=10_000_000))
