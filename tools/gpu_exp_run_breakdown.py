"""Where the wall time of Input.run(1e7) goes, run after run in one process (device sampler following
PCG64, then the host sampler): every Context method wrapped with a timer."""
import os, sys, io, contextlib, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, hip_api
spent, calls = collections.OrderedDict(), collections.Counter()
for name in [n for n in dir(hip_api.Context) if not n.startswith('_') and callable(getattr(hip_api.Context, n))]:
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
ctx = hip_api.Context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
for rep, kw in enumerate([dict(sampler='device', generator='pcg64')]*3 + [dict()]*3):
    inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
    spent.clear(); calls.clear()
    free0 = ctx.mem_info()[0]
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        inputs.run(n, seed=7, context=ctx, **kw)
    dt = time.perf_counter() - t0
    stores = {id(o._store): o._store for o in inputs._catalogue if o._store is not None}
    print(f'run {rep} {kw}: {dt*1e3:.0f} ms, {len(stores)} stores of', ' '.join('%.1f' % (s.nbytes/1e9) for s in stores.values()), 'GB; free before %.0f GB' % (free0/1e9))
    for k, v in sorted(spent.items(), key=lambda kv: -kv[1])[:8]:
        print(f'   {k:28s} {calls[k]:4d} calls {v*1e3:8.1f} ms')
    for s in stores.values():
        s.free()
