"""LOSResult.simulate_data_from_inputs over the catalogue of Input.run(1e6): wall time by part."""
import cProfile, io, os, pstats, sys, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
from nexoclom_amd import Input
from nexoclom_amd.LOSResult import LOSResult, SpacecraftData
from bench_kernels import synthetic_orbit
n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
with contextlib.redirect_stdout(io.StringIO()):
    inputs.run(n, seed=7)
pos, look = synthetic_orbit(S)
sc = SpacecraftData(pos[:, 0], pos[:, 1], pos[:, 2], look[:, 0], look[:, 1], look[:, 2])
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    los = LOSResult(sc, inputs, dphi=np.radians(1.0))
    los.simulate_data_from_inputs(sc)
pr.disable()
print(f'LOSResult over {len(inputs._catalogue)} Outputs x {S} spectra: {time.time()-t0:.2f} s; '
      f'radiance sum {float(np.sum(los.radiance)):.6e}, pairs {int(np.sum(los.npackets_los))}')
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(12)
print(s.getvalue()[:2500])
