"""HIP kernels, through the C ABI, DIRECTLY against the golden vectors that oracle/make_golden.py
produced from the reference's own rk5.py / state.py / histogram.py (tests/golden/g1..g5).  No
oracle call in this module: the driver's GPU record is then self-sufficient -- reference output
in, HIP output compared, with the tolerances of tests/test_oracle_golden.py (the fixtures were
generated with NumPy's pow/exp/log, which are defined to 1 ulp; everything integer-valued --
active steps, alive counts, packet counts per pixel -- must match exactly)."""
import os

import numpy as np
import pytest

from nexoclom_amd.Output import n_output_steps
from tests import helpers as H

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def test_state_against_the_reference_vectors(ctx):
    g = load('g1_state.npz')
    X = g['X']
    for k, (grav, rad, life) in enumerate(g['cfgs']):
        f = H.mercury_forces('Na', 1.3, bool(grav), bool(rad), float(life))
        H.set_ctx_forces(ctx, f)
        a, i = ctx.state(X[:, 1], X[:, 2], X[:, 3], X[:, 5])
        np.testing.assert_allclose(a, g[f'accel{k}'], rtol=2e-15, atol=1e-25)
        assert np.array_equal(i, g[f'ioniz{k}'])


@pytest.mark.parametrize('sp,taa', [('Na', 1.3), ('Ca', 0.0), ('Mg', 3.14)])
def test_rk5_step_against_the_reference_vectors(ctx, sp, taa):
    g = load('g2_rk5.npz')
    f = H.mercury_forces(sp, taa)
    H.set_ctx_forces(ctx, f)
    X, h = g[f'{sp}_X'], g[f'{sp}_h']
    r, d = ctx.rk5_step(X, h, want_delta=True)
    np.testing.assert_allclose(r, g[f'{sp}_result'], rtol=1e-13, atol=1e-20)
    np.testing.assert_allclose(d, g[f'{sp}_delta'], rtol=1e-9, atol=1e-22)
    r30, none = ctx.rk5_step(X, 30.0)
    assert none is None
    np.testing.assert_allclose(r30, g[f'{sp}_result30'], rtol=1e-13, atol=1e-20)
    assert np.array_equal(r30[:, 0], X[:, 0] - 30.)          # the time column is exact: t - h


@pytest.mark.parametrize('name,forces', [('grav', ('Na', 3.14, True, False, 0.0)),
                                         ('na', ('Na', 1.3, True, True, 0.0))])
def test_constant_driver_against_the_reference_vectors(ctx, name, forces):
    g = load('g3_const.npz')
    f = H.mercury_forces(*forces)
    H.set_ctx_forces(ctx, f)
    ctx.set_bounce(None); ctx.set_bodies(None)
    X0 = g[f'{name}_X0']
    endtime, step, edge = g[f'{name}_params']
    nsteps, n_iter = n_output_steps(endtime, step)
    ctx.upload_packets(X0)
    # lock-step kernel with the full trajectory ...
    res = ctx.integrate_const(step, n_iter, edge, nrec=nsteps, want_final=True, want_steps=True)
    assert ctx.counters()['particle_steps'] == int(g[f'{name}_work'])
    assert np.array_equal(res['steps'], g[f'{name}_steps'])
    np.testing.assert_allclose(res['final'], g[f'{name}_final'], rtol=1e-9, atol=1e-13)
    tr = res['traj'].transpose(2, 0, 1)
    assert np.array_equal((tr[:, 7, :] > 0).sum(axis=0), g[f'{name}_alive_per_step'])
    np.testing.assert_allclose(tr[:, 7, :].sum(axis=0), g[f'{name}_fracsum_per_step'], rtol=1e-10)
    np.testing.assert_allclose(tr[g[f'{name}_traj_ids']], g[f'{name}_traj'], rtol=1e-9, atol=1e-13)
    # ... and the persistent lane-refill kernel: same final records, same step counts
    res2 = ctx.integrate_const(step, n_iter, edge, want_final=True, want_steps=True)
    assert ctx.counters()['particle_steps'] == int(g[f'{name}_work'])
    assert np.array_equal(res2['steps'], g[f'{name}_steps'])
    assert np.array_equal(res2['final'], res['final'])
    # compact rows = the frac > 0 records, packet-major
    rows = ctx.integrate_const_rows(step, n_iter, edge)
    live = tr[:, 7, :] > 0
    assert np.array_equal(rows['lengths'], live.sum(axis=1))
    assert np.array_equal(rows['rows'][7], tr[:, 7, :][live])


@pytest.mark.parametrize('quantity', ['radiance', 'column'])
@pytest.mark.parametrize('downcast', [False, True])
def test_fused_image_against_the_reference_vectors(ctx, quantity, downcast):
    g = load('g3_const.npz')
    f = H.mercury_forces('Na', 1.3)
    H.set_ctx_forces(ctx, f)
    ctx.set_bounce(None); ctx.set_bodies(None)
    X0 = g['na_X0']
    endtime, step, edge = g['na_params']
    nsteps, n_iter = n_output_steps(endtime, step)
    im = H.image_setup(f, quantity, dims=(64, 64))
    ctx.set_image(im['M'], f.vrplanet, im['apix'], quantity, im['xedges'], im['zedges'],
                  im['g_tables'], downcast_f32=downcast)
    ctx.upload_packets(X0)
    ctx.integrate_const(step, n_iter, edge, image=True)
    image, counts = ctx.image_download()
    tag = f'na_{quantity}_{"f32" if downcast else "f64"}'
    assert np.array_equal(counts, g[tag + '_counts'].astype(np.uint64))     # bit-exact counts
    np.testing.assert_allclose(image, g[tag + '_image'], rtol=1e-6, atol=0)   # north_star bar
    np.testing.assert_allclose(image, g[tag + '_image'], rtol=1e-9, atol=0)   # what we get
    # two-stage data flow (stored samples -> image kernel) gives the same image
    res = ctx.integrate_const(step, n_iter, edge, nrec=nsteps)
    tr = res['traj']                                       # (8, nsteps, N)
    live = tr[7] > 0
    cols = [tr[c][live] for c in (1, 2, 3, 5, 7)]
    if downcast:
        cols = [c.astype(np.float32).astype(np.float64) for c in cols]
    ctx.set_image(im['M'], f.vrplanet, im['apix'], quantity, im['xedges'], im['zedges'],
                  im['g_tables'], downcast_f32=False)
    ctx.image_accumulate(*cols)
    image2, counts2 = ctx.image_download()
    assert np.array_equal(counts2, g[tag + '_counts'].astype(np.uint64))
    np.testing.assert_allclose(image2, g[tag + '_image'], rtol=1e-9, atol=0)


@pytest.mark.parametrize('quantity', ['radiance', 'column'])
def test_bench_workload_against_g8_from_the_reference_rk5(ctx, quantity):
    """g8_const20k.npz: 20 000 packets of the bench workload through the reference's OWN rk5.py /
    state.py / Histogram2d (oracle/make_golden.py), BASELINE's 512 x 512 image of the float32
    samples.  The kernels fuse the tableau terms (one rounding where NumPy has two), so this --
    not the bit-exact comparison with the C checker -- is what ties their counts to the
    reference: every packet's step count and every pixel's packet count exactly, through the
    fused pass AND through stored rows -> LDS tiles; brightness to 1e-9 (north_star: 1e-6)."""
    g = load('g8_const20k.npz')
    n, seed, endtime, step, edge = g['params']
    f = H.mercury_forces('Na', 1.3)
    H.set_ctx_forces(ctx, f)
    ctx.set_bounce(None); ctx.set_bodies(None)
    X0 = H.sample_x0(int(n), int(seed), endtime)
    nsteps, n_iter = n_output_steps(endtime, step)
    im = H.image_setup(f, quantity, dims=(512, 512))
    ref_cnt = np.zeros(512*512, dtype=np.uint64)
    ref_cnt[g['count_pix']] = g['count_val']

    def check(image, counts):
        assert np.array_equal(counts.ravel(), ref_cnt)
        np.testing.assert_allclose(image.sum(axis=1), g[quantity + '_rowsum'], rtol=1e-9)
        np.testing.assert_allclose(image.sum(axis=0), g[quantity + '_colsum'], rtol=1e-9)
        np.testing.assert_allclose(image.ravel()[::16], g[quantity + '_every16'], rtol=1e-9)
    # the fused pass
    ctx.set_image(im['M'], f.vrplanet, im['apix'], quantity, im['xedges'], im['zedges'],
                  im['g_tables'], downcast_f32=True)
    ctx.upload_packets(X0)
    res = ctx.integrate_const(step, n_iter, edge, image=True, want_steps=True)
    assert ctx.counters()['particle_steps'] == int(g['work'])
    assert np.array_equal(res['steps'], g['steps'].astype(res['steps'].dtype))
    check(*ctx.image_download())
    # stored float32 rows (what save() keeps) -> k_image_bin + k_image_tiles, and -> k_image
    ctx.upload_packets(X0)
    store = ctx.integrate_const_rows(step, n_iter, edge, narrow=True, resident=True)['store']
    alive = g['alive_per_step']
    assert store.total == int(alive.sum())
    try:
        for mode in ('tiles', 'atomics'):
            ctx.set_image(im['M'], f.vrplanet, im['apix'], quantity, im['xedges'], im['zedges'],
                          im['g_tables'], downcast_f32=False)
            ctx.image_mode(mode)
            ctx.image_accumulate_rows(store)
            check(*ctx.image_download())
        # the stored rows themselves: records alive and sum(frac) per step (float32 values)
        rows, _ = store.download()
        k = np.rint((endtime - rows[0].astype(np.float64))/step).astype(np.int64)
        assert np.array_equal(np.bincount(k, minlength=nsteps), alive)
        np.testing.assert_allclose(np.bincount(k, weights=rows[7].astype(np.float64),
                                               minlength=nsteps),
                                   g['fracsum_per_step'], rtol=1e-6)
    finally:
        ctx.image_mode('auto')
        store.free()


def test_variable_driver_against_the_reference_vectors(ctx):
    g = load('g4_var.npz')
    f = H.mercury_forces('Na', 1.3)
    H.set_ctx_forces(ctx, f)
    ctx.set_bounce(None); ctx.set_bodies(None)
    res, edge = g['params']
    ctx.upload_packets(g['X0'])
    fin, hs = ctx.integrate_var(float(res), float(edge))
    ctr = ctx.counters()
    assert ctr['particle_steps'] == int(g['work'])
    assert ctr['unfinished'] == ctr['bad_step'] == ctr['nonfinite'] == ctr['neg_frac'] == 0
    np.testing.assert_allclose(fin, g['final'], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(hs, g['step_size'], rtol=1e-9)


def test_histogram_edge_cases_against_the_reference_vectors(ctx):
    g = load('g5_hist.npz')
    px, pz, w = g['px'], g['pz'], g['w']
    edges = np.linspace(-4, 4, 513)
    # identity rotation, column weights = frac, Apix = 1: the image IS the weighted histogram
    ctx.set_image(np.eye(3), 0.0, 1.0, 'column', edges, edges, [])
    # y = -1 keeps every sample in view of the observer (ModelImage.py:252-254)
    ctx.image_accumulate(px, -np.ones_like(px), pz, np.zeros_like(px), w)
    img, cnt = ctx.image_download()
    i, j = g['nz_i'], g['nz_j']
    assert cnt.sum() == g['counts'].sum()
    assert np.array_equal(cnt[i, j].astype(float), g['counts'])
    np.testing.assert_allclose(img[i, j], g['weights'], rtol=1e-13)
    assert cnt[511, :].sum() > 0          # samples == right-most edge land in the last bin
    ref, _, _ = np.histogram2d(px, pz, bins=[512, 512], range=[[-4, 4], [-4, 4]])
    assert np.array_equal(cnt.astype(float), ref)


def test_variable_driver_against_g9_from_the_reference_rk5(ctx):
    """g9_var2000.npz: 2000 packets of the bench workload at random ages through the reference's own
    rk5.py in the adaptive driver (Output.py:221-366).  The kernels' error estimate uses the fused
    tableau terms too, so its accept / reject decisions could in principle differ from the
    reference's: the TOTAL number of attempts must be the reference's, every stored step size agree
    to 1e-9 and every final state to 1e-8."""
    g = load('g9_var2000.npz')
    f = H.mercury_forces('Na', 1.3)
    H.set_ctx_forces(ctx, f)
    ctx.set_bounce(None); ctx.set_bodies(None)
    res, edge = g['params']
    ctx.upload_packets(g['X0'])
    fin, hs = ctx.integrate_var(float(res), float(edge))
    ctr = ctx.counters()
    assert ctr['particle_steps'] == int(g['work']) > 5e5
    assert ctr['unfinished'] == ctr['bad_step'] == ctr['nonfinite'] == ctr['neg_frac'] == 0
    np.testing.assert_allclose(hs, g['step_size'], rtol=1e-9)
    np.testing.assert_allclose(fin, g['final'], rtol=1e-8, atol=1e-12)
