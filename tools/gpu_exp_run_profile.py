"""cProfile of Input.run(1e6) (host-sampled and pcg64) on the GPU box: where the host time goes."""
import contextlib, cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, hip_api
infile = os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input')
ctx = hip_api.Context(0)
for kw in ({}, dict(sampler='device', generator='pcg64')):
    for rep in range(3):
        inputs = Input(infile)
        pr = cProfile.Profile() if rep == 2 else None
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            if pr: pr.enable()
            inputs.run(1e6, seed=7, context=ctx, **kw)
            if pr: pr.disable()
        print(kw, f'Input.run(1e6) {time.time() - t0:.3f} s', flush=True)
        if pr:
            out = io.StringIO()
            pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(30)
            print('\n'.join(l for l in out.getvalue().splitlines()[4:42]))
        for o in inputs._catalogue:
            if o._store is not None:
                o._store.free()
        del inputs
