"""The image of stored samples through LDS-privatised tiles (k_image_bin + k_image_tiles) against
np.histogram2d and against the atomic path (k_image): identical packet counts, weight sums equal to
the order of fp64 additions.  What it replaces: np.histogram2d's bincount, math/histogram.py:34,
called from data_simulation/ModelImage.py:267-269."""
import os

import numpy as np
import pytest

from oracle import np_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture
def tiles(ctx):
    """Leaves the context in its default mode whatever the test did."""
    yield ctx
    ctx.image_mode('auto')


def _set(ctx, f, im, quantity):
    ctx.set_image(im['M'], f.vrplanet, im['apix'], quantity, im['xedges'], im['zedges'],
                  im['g_tables'])


# (dims, tile_pixels, slab_samples): 32 row-interleaved tiles of a 512^2 image; a non-power-of-two
# image in 4 tiles; the same in 32 tiles of 8 rows and five slabs; a single tile; the reference's
# DEFAULT image, 800 x 800 (ModelImage.py:53): 128 tiles of 7 rows, 64-entry chunks; 1024^2: 128
# tiles of 8 rows, every pixel of every LDS tile in use; 640 x 700: 64 tiles, 128-entry chunks;
# 800 x 800 in three slabs
@pytest.mark.parametrize('dims,tile_pixels,slab', [((512, 512), 0, 0), ((200, 120), 0, 0),
                                                   ((200, 120), 1024, 70001), ((64, 64), 0, 0),
                                                   ((800, 800), 0, 0), ((1024, 1024), 0, 0),
                                                   ((640, 700), 0, 0), ((800, 800), 0, 100003)])
@pytest.mark.parametrize('quantity', ['radiance', 'column'])
@pytest.mark.parametrize('f32', [False, True])
def test_tiled_image_equals_numpy_histogram_and_the_atomic_path(tiles, dims, tile_pixels, slab,
                                                                quantity, f32):
    """f32: the samples are float32 values (as save() stores them); pass 2 then forms the weights
    (k_image_bin<DEFER> / k_image_tiles<WEIGH>).  Otherwise pass 1 does, in fp64."""
    ctx = tiles
    f = H.mercury_forces('Na', 1.3)
    rng = np.random.default_rng(7)
    p = 300001
    X = H.random_cloud(p, 22)
    if f32:
        X = X.astype(np.float32).astype(np.float64)
    x, y, z, vy, frac = (np.ascontiguousarray(X[:, c]) for c in (1, 2, 3, 5, 7))
    im = H.image_setup(f, quantity, dims=dims, width=(8., 8. * dims[1] / dims[0]))
    # samples exactly on edges (the right-most included), outside, non-finite
    nxe, nze = len(im['xedges']), len(im['zedges'])
    x[:nxe] = im['xedges']; z[:nze] = im['zedges'][::-1]
    x[600:610] = im['xedges'][-1]; z[600:610] = rng.uniform(im['zedges'][0], im['zedges'][-1], 10)
    x[700:710] = im['xedges'][-1]*(1 + 1e-7)
    x[800:803] = [np.nan, np.inf, -np.inf]
    frac[902:904] = [np.inf, np.nan]                                  # non-finite weights (:170)
    if quantity == 'radiance':
        vy[900:902] = [np.nan, np.inf]
    if f32:
        x, y, z, vy, frac = (c.astype(np.float32) for c in (x, y, z, vy, frac))
    _set(ctx, f, im, quantity)
    ctx.image_mode('tiles', tile_pixels, slab)
    ctx.image_accumulate(x, y, z, vy, frac)
    image, counts = ctx.image_download()
    ctr = ctx.counters()
    ctx.image_clear()
    ctx.image_mode('atomics')
    ctx.image_accumulate(x, y, z, vy, frac)
    image1, counts1 = ctx.image_download()
    assert ctr == ctx.counters() and ctr['samples'] == p
    assert ctr['samples_binned'] == int(counts.sum()) > 10000
    assert np.array_equal(counts, counts1)
    np.testing.assert_allclose(image, image1, rtol=1e-12, atol=0)
    assert ctr['nonfinite'] >= 1
    x, y, z, vy, frac = (c.astype(np.float64) for c in (x, y, z, vy, frac))
    ok = np.isfinite(x) & np.isfinite(vy) & np.isfinite(frac)
    ref_img, ref_cnt, _, _ = O.create_image(x[ok], y[ok], z[ok], vy[ok], frac[ok], f.vrplanet,
                                            im['M'], quantity, im['g_tables'], im['dims'],
                                            im['xrange'], im['zrange'], im['apix'], matmul=False)
    assert np.array_equal(counts, ref_cnt.astype(np.uint64))
    np.testing.assert_allclose(image, ref_img, rtol=1e-12, atol=0)


def test_tiled_image_when_every_sample_lands_in_one_tile(tiles):
    """All samples in one image row: one staging block takes every entry of a trip, overflows and
    is emptied several times per trip (the retry rounds of k_image_bin); float32 samples."""
    ctx = tiles
    f = H.mercury_forces('Na', 1.3)
    p = 150000
    X = H.random_cloud(p, 23).astype(np.float32)
    cols = [np.ascontiguousarray(X[:, c]) for c in (1, 2, 3, 5, 7)]
    cols[0][:] = np.float32(0.3)                  # one x: one image row, one tile
    im = H.image_setup(f, 'column', dims=(512, 512))
    _set(ctx, f, im, 'column')
    ctx.image_mode('tiles')
    ctx.image_accumulate(*cols)
    image, counts = ctx.image_download()
    assert ctx.counters()['samples_binned'] == int(counts.sum()) > 100000
    assert np.count_nonzero(counts.sum(axis=1)) == 1
    ctx.image_clear()
    ctx.image_mode('atomics')
    ctx.image_accumulate(*cols)
    image1, counts1 = ctx.image_download()
    assert np.array_equal(counts, counts1)
    np.testing.assert_allclose(image, image1, rtol=1e-12, atol=0)


def test_tiled_image_against_the_reference_histogram_vectors(tiles):
    """g5_hist: np.histogram2d's own answers for samples on and around the 512^2 edges."""
    ctx = tiles
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g5_hist.npz'),
                allow_pickle=False)
    px, pz, w = g['px'], g['pz'], g['w']
    edges = np.linspace(-4, 4, 513)
    ctx.set_image(np.eye(3), 0.0, 1.0, 'column', edges, edges, [])
    ctx.image_mode('tiles')
    ctx.image_accumulate(px, -np.ones_like(px), pz, np.zeros_like(px), w)
    img, cnt = ctx.image_download()
    i, j = g['nz_i'], g['nz_j']
    assert np.array_equal(cnt[i, j].astype(float), g['counts'])
    np.testing.assert_allclose(img[i, j], g['weights'], rtol=1e-13)
    ref, _, _ = np.histogram2d(px, pz, bins=[512, 512], range=[[-4, 4], [-4, 4]])
    assert np.array_equal(cnt.astype(float), ref)


@pytest.mark.parametrize('dims', [(800, 800), (1024, 1024)])
def test_large_tiled_image_counts_equal_numpy_histogram2d(tiles, dims):
    """np.histogram2d itself (math/histogram.py:34) at the reference's default image size and at
    1024^2, on 2e6 float32 samples with every edge value among them: exact counts."""
    ctx = tiles
    rng = np.random.default_rng(11)
    p = 2_000_000
    edges_x, edges_z = np.linspace(-4, 4, dims[0] + 1), np.linspace(-4, 4, dims[1] + 1)
    px = rng.normal(0, 1.7, p).astype(np.float32)
    pz = rng.normal(0, 1.7, p).astype(np.float32)
    px[:dims[0] + 1] = edges_x.astype(np.float32)
    pz[dims[0] + 1:dims[0] + dims[1] + 2] = edges_z.astype(np.float32)
    w = rng.uniform(0.1, 1, p).astype(np.float32)
    ctx.set_image(np.eye(3), 0.0, 1.0, 'column', edges_x, edges_z, [])
    ctx.image_mode('tiles')
    ctx.image_accumulate(px, -np.ones_like(px), pz, np.zeros_like(px), w)
    img, cnt = ctx.image_download()
    x64, z64 = px.astype(np.float64), pz.astype(np.float64)
    ref, _, _ = np.histogram2d(x64, z64, bins=list(dims), range=[[-4, 4], [-4, 4]])
    refw, _, _ = np.histogram2d(x64, z64, bins=list(dims), range=[[-4, 4], [-4, 4]],
                                weights=w.astype(np.float64))
    assert np.array_equal(cnt.astype(float), ref) and ref.sum() > 1.8e6
    np.testing.assert_allclose(img, refw, rtol=1e-12, atol=0)


def test_tiled_image_refuses_what_it_cannot_hold(tiles):
    ctx = tiles
    f = H.mercury_forces('Na', 1.3)
    im = H.image_setup(f, 'column', dims=(1032, 1024))       # one row beyond 128 tiles of 8
    _set(ctx, f, im, 'column')
    ctx.image_mode('tiles')
    from nexoclom_amd.hip_api import HipError
    x = np.zeros(10)
    with pytest.raises(HipError, match='tiled'):
        ctx.image_accumulate(x, x, x, x, x)
    ctx.image_mode('auto')                        # by size: such an image stays with the atomics
    ctx.image_accumulate(x, x - 1, x, x, x + 1)
    assert ctx.image_download()[1].sum() == 10
    with pytest.raises(HipError):
        ctx.image_mode(3)


@pytest.mark.parametrize('narrow', [False, True])
def test_resident_rows_through_the_tiles(tiles, narrow):
    """nxc_image_accumulate_rows over a row range of a resident store (64-bit and float32 columns,
    an odd first row), several slabs of tiles against the atomic path."""
    ctx = tiles
    f = H.mercury_forces('Na', 1.3)
    endtime, step = 9000., 30.
    X0 = H.sample_x0(5000, 34, endtime)
    nsteps, n_iter = O.n_output_steps(endtime, step)
    H.set_ctx_forces(ctx, f)
    ctx.set_bounce(None); ctx.set_bodies(None)
    ctx.upload_packets(X0)
    store = ctx.integrate_const_rows(step, n_iter, 8.0, narrow=narrow, resident=True)['store']
    im = H.image_setup(f, 'radiance', dims=(96, 96))
    a, b = 777, store.total - 999
    out = {}
    for mode, args in (('atomics', ()), ('tiles', (1024, 100003))):
        _set(ctx, f, im, 'radiance')
        ctx.image_mode(mode, *args)
        ctx.image_accumulate_rows(store, a, b - a)
        out[mode] = ctx.image_download() + (ctx.counters(),)
    store.free()
    assert out['tiles'][2] == out['atomics'][2] and out['tiles'][2]['samples'] == b - a
    assert out['tiles'][1].sum() > 1e5 and np.array_equal(out['tiles'][1], out['atomics'][1])
    np.testing.assert_allclose(out['tiles'][0], out['atomics'][0], rtol=1e-12, atol=0)


def test_produce_image_at_full_size_both_ways_and_against_the_fused_pass(tiles):
    """The two-stage flow at the size of BASELINE configs[1]: Input.run(1e6) leaves 1.3e8 float32
    rows in HBM; produce_image over them through the tiles (the default at this size) and through
    k_image gives the same packet count in every pixel -- and the count image of the FUSED pass
    over the same seeded packets (ModelImage(npackets=...), the path pinned against the C oracle
    at this size in test_gpu_edge_fullsize.py)."""
    import contextlib
    import io
    import nexoclom_amd
    from nexoclom_amd import Input, ModelImage
    ctx = tiles
    inputs = Input(os.path.join(os.path.dirname(nexoclom_amd.__file__), 'inputfiles',
                                'Na.mercury.bench.input'))
    params = {'quantity': 'radiance', 'dims': '512,512'}
    with contextlib.redirect_stdout(io.StringIO()):
        inputs.run(1e6, seed=5, context=ctx, sampler='device', generator='pcg64')
        outs = inputs._catalogue
        images = {}
        for mode in ('auto', 'atomics', 'tiles'):
            ctx.image_mode(mode)
            images[mode] = inputs.produce_image(params, context=ctx)
        ctx.image_mode('auto')
        fused = ModelImage(inputs, params, npackets=len(outs)*len(outs[0]),
                           packs_per_it=len(outs[0]), seed=5, sampler='device', generator='pcg64',
                           context=ctx)
    rows = sum(o._nrows for o in outs)
    assert rows > 1.2e8
    a, t = images['atomics'], images['tiles']
    assert t.packet_image.sum() > 6e7
    assert np.array_equal(t.packet_image, a.packet_image)
    assert np.array_equal(t.packet_image, images['auto'].packet_image)
    assert np.array_equal(t.packet_image, fused.packet_image)
    np.testing.assert_allclose(t.image, a.image, rtol=1e-12, atol=0)
    np.testing.assert_allclose(t.image, fused.image, rtol=1e-11, atol=0)
    assert t.totalsource == a.totalsource == fused.totalsource
