"""cProfile of Input.run(1e6, sampler='device', generator='pcg64') (three repetitions)."""
import contextlib, cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, hip_api
ctx = hip_api.Context(0)
infile = os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input')
for rep in range(3):
    inputs = Input(infile)
    prof = cProfile.Profile()
    t0 = time.time()
    prof.enable()
    with contextlib.redirect_stdout(io.StringIO()):
        inputs.run(1e6, seed=7, context=ctx, sampler='device', generator='pcg64')
    prof.disable()
    print(f'rep {rep}: {time.time() - t0:.3f} s', flush=True)
    if rep:
        pstats.Stats(prof).sort_stats('tottime').print_stats(8)
    del inputs
