"""cProfile of Input.run(N, sampler='device', generator='pcg64') (default N = 1e6; three repetitions)."""
import contextlib, cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, hip_api
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
ctx = hip_api.Context(0)
infile = os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input')
for rep in range(3):
    inputs = Input(infile)
    prof = cProfile.Profile()
    t0 = time.time()
    prof.enable()
    with contextlib.redirect_stdout(io.StringIO()):
        inputs.run(n, seed=7, context=ctx, sampler='device', generator='pcg64')
    prof.disable()
    print(f'rep {rep}: {time.time() - t0:.3f} s', flush=True)
    if rep == 2:
        pstats.Stats(prof).sort_stats('tottime').print_stats(14)
    del inputs
