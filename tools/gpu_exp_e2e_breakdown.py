"""Where the wall time of ModelImage(npackets=1e8, sampler='device') goes: every Context method
is wrapped with a timer (each call synchronises, so host time = device time + overhead)."""
import os, sys, io, contextlib, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, ModelImage, hip_api
spent = collections.OrderedDict()
calls = collections.Counter()
for name in ('sample_packets', 'integrate_const', 'counters', 'set_forces', 'set_image', 'set_bounce',
             'set_bodies', 'set_first_index', 'image_download', 'image_clear', 'upload_soa'):
    def wrap(fn, name=name):
        def timed(self, *a, **k):
            t0 = time.perf_counter()
            try:
                return fn(self, *a, **k)
            finally:
                spent[name] = spent.get(name, 0.) + time.perf_counter() - t0
                calls[name] += 1
        return timed
    setattr(hip_api.Context, name, wrap(getattr(hip_api.Context, name)))
t0 = time.perf_counter()
ctx = hip_api.Context(0)
t_ctx = time.perf_counter() - t0
bench = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
params = {'quantity': 'radiance', 'dims': '512,512'}
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
for rep in range(2):
    spent.clear(); calls.clear()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        img = ModelImage(bench, params, npackets=n, seed=1, context=ctx, sampler='device')
    dt = time.perf_counter() - t0
    print(f'pass {rep}: {dt*1e3:.0f} ms wall for {n} packets (context creation before it: {t_ctx*1e3:.0f} ms)')
    for k, v in spent.items():
        print(f'   {k:18s} {calls[k]:3d} calls {v*1e3:8.1f} ms')
    print(f'   {"host (rest)":18s}           {(dt - sum(spent.values()))*1e3:8.1f} ms')
