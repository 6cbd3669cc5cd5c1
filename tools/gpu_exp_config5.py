"""BASELINE config 5 on one GPU: Na from Io, Io + Europa gravity, torus loss, fused image."""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import nexoclom_amd
from nexoclom_amd import Input, ModelImage, hip_api

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
infile = os.path.join(os.path.dirname(nexoclom_amd.__file__), 'inputfiles', 'Na.io.torus.input')
inputs = Input(infile)
ctx = hip_api.Context(0)
params = {'quantity': 'radiance', 'dims': '512,512', 'width': '30,30'}
for rep in range(2):
    t0 = time.time()
    img = ModelImage(inputs, params, npackets=n, seed=11, context=ctx)
    wall = time.time() - t0
    ms = ctx.last_kernel_ms()
    ps = img.counters['particle_steps']
    print(f'n={n} wall={wall:.2f}s kernel={ms:.1f}ms particle_steps={ps:.3e} '
          f'-> {ps/ms*1e3:.3e} p.s/s  binned={img.counters["samples_binned"]:.3e} '
          f'mean steps/packet={ps/n:.0f}', flush=True)
print('image max', img.image.max(), 'nonzero pixels', (img.packet_image > 0).sum())
