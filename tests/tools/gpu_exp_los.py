"""Timing of the LOS cone kernel and the stand-alone image kernel on stored samples."""
import os, sys, time, io, contextlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, Output, ModelImage, LOSResult, SpacecraftData, hip_api
from nexoclom_amd.LOSResult import los_geometry, arccos_threshold
from tests.test_gpu_api import _orbit

inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
ctx = hip_api.Context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
with contextlib.redirect_stdout(io.StringIO()):
    out = Output(inputs, n, seed=1, context=ctx, save=False)
X = out.X[out.X.frac > 0]
P = len(X)
print('samples', P)
pos, look = _orbit(S)
sc = SpacecraftData(pos[:, 0], pos[:, 1], pos[:, 2], look[:, 0], look[:, 1], look[:, 2])
los = LOSResult(sc, inputs, dphi=np.radians(1.0), context=ctx)
dist, lengths, ladder = los_geometry(sc.data, 25., los.dphi)
scarr = np.stack([sc.data.x, sc.data.y, sc.data.z, sc.data.xbore, sc.data.ybore, sc.data.zbore, dist, lengths.astype(float)])
args = (los.dphi, np.sin(los.dphi), np.sin(2*los.dphi), arccos_threshold(los.dphi), float(out.vrplanet)/out.unit_km,
        out.unit_km*1e5, los.g_tables(float(out.aplanet)), ladder, scarr,
        X.x.values, X.y.values, X.z.values, X.vy.values, X.frac.values)
for _ in range(3):
    t0 = time.time(); r = ctx.los_accumulate(*args); t1 = time.time()
    ms = ctx.last_kernel_ms()
    print(f'LOS: {P} samples x {S} spectra: kernel {ms:.2f} ms -> {P*S/ms/1e6:.1f} G pairs/s (call {1e3*(t1-t0):.0f} ms), pairs in cones {int(r["npackets"].sum())}')
img = ModelImage(inputs, {'quantity': 'radiance', 'dims': '512,512'}, context=ctx)
img._set_image(ctx, float(out.aplanet), float(out.vrplanet)/out.unit_km, False)
for _ in range(3):
    ctx.image_clear(); ctx.image_accumulate(X.x.values, X.y.values, X.z.values, X.vy.values, X.frac.values)
    ms = ctx.last_kernel_ms()
    print(f'k_image: {P} samples: kernel {ms:.3f} ms -> {P/ms/1e6:.2f} G samples/s, {40*P/ms/1e6:.0f} GB/s algorithmic')
