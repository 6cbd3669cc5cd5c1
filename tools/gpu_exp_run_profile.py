"""cProfile of Input.run(n) (host-sampled, then pcg64, in one process) on the GPU box: where the
host time goes.  python tools/gpu_exp_run_profile.py [n]"""
import contextlib, cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, hip_api
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
infile = os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input')
ctx = hip_api.Context(0)
for rep in range(2):
    for kw in ({}, dict(sampler='device', generator='pcg64')):
        inputs = Input(infile)
        pr = cProfile.Profile()
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            pr.enable()
            inputs.run(n, seed=7, context=ctx, **kw)
            pr.disable()
        print(kw, f'Input.run({n}) {time.time() - t0:.3f} s', flush=True)
        out = io.StringIO()
        pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(12)
        print('\n'.join(l[:150] for l in out.getvalue().splitlines()[4:22]))
        t0 = time.time()
        for o in inputs._catalogue:
            if o._store is not None:
                o._store.free()
        del inputs
        print(f'   freeing the stores {time.time() - t0:.3f} s', flush=True)
