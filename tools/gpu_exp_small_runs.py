"""Wall time of the drop-in flow at SMALL sizes (BASELINE configs[0]: Ca.isotropic.flat.input, 1e4
packets; and the bench inputfile at 1e4 / 1e5), where launch and set-up overheads are all there is:
Context creation, Input(), run(), produce_image(), a second run."""
import contextlib, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
from nexoclom_amd import Input, hip_api
t_import = time.perf_counter() - t0
t0 = time.perf_counter(); ctx = hip_api.Context(0); t_ctx = time.perf_counter() - t0
print(json.dumps({'import_s': t_import, 'context_s': t_ctx}))
for name, n in (('Ca.isotropic.flat.input', 10_000), ('Na.mercury.bench.input', 10_000), ('Na.mercury.bench.input', 100_000)):
    for rep in range(3):
        t0 = time.perf_counter()
        inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', name))
        t1 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            inputs.run(n, seed=5, context=ctx)
        t2 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            img = inputs.produce_image({'quantity': 'column', 'dims': '512,512'}, context=ctx)
        t3 = time.perf_counter()
        print(json.dumps({'inputfile': name, 'packets': n, 'rep': rep, 'Input_ms': (t1 - t0)*1e3, 'run_ms': (t2 - t1)*1e3,
                          'produce_image_ms': (t3 - t2)*1e3, 'binned': float(img.packet_image.sum())}), flush=True)
        for o in inputs._catalogue:
            if o._store is not None: o._store.free()
