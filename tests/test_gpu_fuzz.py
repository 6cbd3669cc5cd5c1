"""Randomised differential test: HIP (C ABI) vs C oracle over random force / step / image
configurations.  Every state column, step count and packet count must match bit for bit."""
import os

import numpy as np
import pytest

from oracle import np_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _random_case(rng):
    species = rng.choice(['Na', 'Ca', 'Mg'])
    taa = float(rng.uniform(0, 2*np.pi))
    gravity = bool(rng.random() < 0.85)
    radpres = bool(rng.random() < 0.8)
    mode = rng.choice(['photo', 'lifetime', 'generic'])
    lifetime = {'photo': 0.0, 'lifetime': float(rng.uniform(500, 20000)),
                'generic': -float(rng.uniform(500, 20000))}[mode]
    f = H.mercury_forces(species, taa, gravity, radpres, lifetime)
    step = float(rng.choice([30.0, 17.3, 45.5, 10.0, 61.0]))
    endtime = float(rng.choice([3000.0, 7000.0, 12345.6, 20000.0]))
    outeredge = float(rng.choice([3.0, 8.0, 25.0, 1e30]))
    quantity = rng.choice(['radiance', 'column'])
    dims = (int(rng.integers(16, 200)), int(rng.integers(16, 200)))
    width = (float(rng.uniform(2, 12)), float(rng.uniform(2, 12)))
    center = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)))
    lines = {'Na': (5891, 5897), 'Ca': (4227,), 'Mg': (2852,)}[species]
    im = H.image_setup(f, quantity, dims=dims, center=center, width=width,
                       sublon=float(rng.uniform(0, 2*np.pi)),
                       sublat=float(rng.uniform(-np.pi/2, np.pi/2)), species=species,
                       wavelengths=lines)
    return f, step, endtime, outeredge, quantity, im, bool(rng.random() < 0.5)


@pytest.mark.parametrize('seed', range(int(os.environ.get('NXC_FUZZ_SEEDS', '16'))))   # soak: more
def test_random_configuration_parity(ctx, coracle, seed):
    rng = np.random.default_rng(1000 + seed)
    f, step, endtime, outeredge, quantity, im, downcast = _random_case(rng)
    n = int(rng.integers(500, 3000))
    X0 = H.sample_x0(n, 77 + seed, endtime, vprob=float(rng.uniform(1.5, 3.5)),
                     delv=float(rng.uniform(0.5, 1.5)))
    nsteps, n_iter = O.n_output_steps(endtime, step)
    n_iter = min(n_iter, nsteps - 1)
    H.set_ctx_forces(ctx, f)
    ctx.set_image(im['M'], f.vrplanet, im['apix'], quantity, im['xedges'], im['zedges'],
                  im['g_tables'], downcast_f32=downcast)
    ctx.upload_packets(X0)
    g = ctx.integrate_const(step, n_iter, outeredge, image=True, want_final=True, want_steps=True)
    image, counts = ctx.image_download()
    ctr = ctx.counters()
    desc = coracle.image_desc(im['M'], f.vrplanet, im['apix'], quantity, im['g_tables'],
                              im['xedges'], im['zedges'], downcast=downcast)
    c = coracle.integrate_const(f, X0, step, n_iter, outeredge, img=desc, threads=4)
    assert np.array_equal(g['steps'], c['steps'])
    assert np.array_equal(g['final'], c['final'])
    assert ctr['particle_steps'] == c['work'] and ctr['nonfinite'] == 0
    assert np.array_equal(counts, c['counts'])
    np.testing.assert_allclose(image, c['image'], rtol=1e-11, atol=0)
    # the lock-step trajectory kernel on a subset
    m = min(n, 400)
    ctx.upload_packets(X0[:m])
    t = ctx.integrate_const(step, n_iter, outeredge, nrec=nsteps, want_final=True)
    ct = coracle.integrate_const(f, X0[:m], step, n_iter, outeredge, nrec=nsteps)
    assert np.array_equal(t['traj'], ct['traj'])
    # the compact rows (what save() keeps, Output.py:523-543): 64-bit, narrowed on the device, and
    # as a resident store with its packet-index column
    live = ct['traj'][7].T > 0                                           # (m, nsteps)
    ctx.upload_packets(X0[:m])
    wide = ctx.integrate_const_rows(step, n_iter, outeredge)
    assert ctx.counters()['unfinished'] == 0
    assert np.array_equal(wide['lengths'], live.sum(1))
    for col in range(8):
        assert np.array_equal(wide['rows'][col], ct['traj'][col].T[live])
    store = ctx.integrate_const_rows(step, n_iter, outeredge, narrow=True, resident=True)['store']
    rows32, index = store.download()
    store.free()
    assert np.array_equal(rows32, wide['rows'].astype(np.float32))
    assert np.array_equal(index, np.repeat(np.arange(m), wide['lengths']))
    # and one adaptive run
    Xv = X0[:m].copy()
    Xv[:, 0] = rng.random(m)*endtime
    ctx.upload_packets(Xv)
    res = float(rng.choice([1e-4, 1e-3, 1e-5]))
    vf, vh = ctx.integrate_var(res, outeredge)
    cf, chs, work, bad = coracle.integrate_var(f, Xv, res, outeredge)
    assert bad == 0 and np.array_equal(vf, cf) and np.array_equal(vh, chs)
