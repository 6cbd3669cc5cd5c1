import cProfile, io, os, pstats, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nexoclom_amd import Input
inputs = Input(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
inputs.options.step_size = 0.
inputs.options.resolution = 1e-4
pr = cProfile.Profile()
t0=time.time(); pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    inputs.run(1e6, seed=7)
t1=time.time()
with contextlib.redirect_stdout(io.StringIO()):
    img = inputs.produce_image({'quantity': 'radiance', 'dims': '512,512'})
pr.disable(); t2=time.time()
print(f'variable: Input.run(1e6) {t1-t0:.2f} s, produce_image {t2-t1:.2f} s, binned {img.packet_image.sum()}')
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:3000])
