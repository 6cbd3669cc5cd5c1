"""Where the time goes in the reference-style two-stage flow (Input.run with materialised
trajectories -> catalogue -> ModelImage from the catalogue) for one reference-sized chunk."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, '.')
import nexoclom_amd
from nexoclom_amd import Input, ModelImage, hip_api
from nexoclom_amd.Output import Output

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 80467
infile = os.path.join(os.path.dirname(nexoclom_amd.__file__), 'inputfiles', 'Na.mercury.bench.input')
inputs = Input(infile)
ctx = hip_api.Context(0)
tmp = tempfile.mkdtemp(dir='gpurun_out')
for savepath in (None, tmp):
    inputs._catalogue.clear()
    inputs.savepath = savepath
    t0 = time.time()
    out = Output(inputs, n, seed=1, context=ctx, save=False)
    t1 = time.time()
    out.save()
    t2 = time.time()
    img = ModelImage(inputs, {'quantity': 'radiance', 'dims': '512,512'}, context=ctx)
    t3 = time.time()
    print(f'n={n} savepath={bool(savepath)}: Output(integrate+frame) {t1-t0:.2f}s '
          f'(kernel {0:.0f}) save {t2-t1:.2f}s ModelImage(catalogue) {t3-t2:.2f}s rows={len(out.X)}', flush=True)
import shutil; shutil.rmtree(tmp)

# the reference's own user flow at 1e6 packets: Input.run (chunked like Input.py:219-222, every
# chunk catalogued with its trajectory) then Input.produce_image
import contextlib, io, cProfile, pstats
for rep in range(2):
    inputs = Input(infile)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        inputs.run(1e6, seed=7, context=ctx)
    t1 = time.time()
    img = inputs.produce_image({'quantity': 'radiance', 'dims': '512,512'}, context=ctx)
    t2 = time.time()
    rows = sum(o._nrows for o in inputs._catalogue)
    print(f'Input.run(1e6): {t1-t0:.3f}s in {len(inputs._catalogue)} chunks, {rows:.3e} rows resident '
          f'in HBM; produce_image {t2-t1:.3f}s; total {t2-t0:.3f}s', flush=True)
    t3 = time.time()
    X = inputs._catalogue[0].X
    t4 = time.time()
    print(f'   first access to one Output.X ({len(X):.3e} rows): {t4-t3:.3f}s', flush=True)
    img_host = img.packet_image.copy()
    del inputs, img, X
for rep in range(2):
    inputs = Input(infile)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        inputs.run(1e6, seed=7, context=ctx, sampler='device', generator='pcg64')
    t1 = time.time()
    img2 = inputs.produce_image({'quantity': 'radiance', 'dims': '512,512'}, context=ctx)
    t2 = time.time()
    print(f"Input.run(1e6, sampler='device', generator='pcg64'): {t1-t0:.3f}s; produce_image "
          f'{t2-t1:.3f}s; total {t2-t0:.3f}s; same packet counts as the host-sampled run: '
          f'{bool((img2.packet_image == img_host).all())}', flush=True)
    del inputs, img2
inputs = Input(infile)
prof = cProfile.Profile()
prof.enable()
with contextlib.redirect_stdout(io.StringIO()):
    inputs.run(1e6, seed=7, context=ctx)
prof.disable()
pstats.Stats(prof).sort_stats('cumulative').print_stats(18)
