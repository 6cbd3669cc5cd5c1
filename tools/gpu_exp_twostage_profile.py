"""cProfile of the reference-style two-stage flow (Input.run(1e6) with catalogued trajectories,
then produce_image): which host functions hold the wall time."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input
n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
inputs.run(n, seed=7)
t1 = time.time()
img = inputs.produce_image({'quantity': 'radiance', 'dims': '512,512'})
pr.disable()
t2 = time.time()
print(f'Input.run({n:g}) {t1-t0:.2f} s, produce_image {t2-t1:.2f} s', file=sys.stderr)
for key in ('cumulative', 'tottime'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print(s.getvalue()[:5000], file=sys.stderr)
