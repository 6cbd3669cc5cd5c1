"""The drop-in boundary on a real GPU: Input / Output / ModelImage used the way a nexoclom user
(and the reference's own tests) use them."""
import contextlib
import io
import os

import numpy as np
import pytest

import nexoclom_amd
from nexoclom_amd import Input, ModelImage, Output
from oracle import np_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
PKG_INPUTS = os.path.join(os.path.dirname(nexoclom_amd.__file__), 'inputfiles')


@pytest.mark.parametrize('run', ['constant', 'variable'])
def test_gravity_energy_conservation(ctx, run):
    """The reference's tests/unit_tests/particle_tracking/test_gravity.py:22-55: packets under
    gravity alone conserve v^2/2 + GM/r along their trajectory (constant and variable step)."""
    inputs = Input(os.path.join(HERE, 'inputfiles', 'Gravity.input'))
    if run == 'variable':
        inputs.options.step_size = 0
        inputs.options.resolution = 0.0001
    else:
        inputs.options.step_size = 30
    inputs.run(1e3, 1e3, overwrite=False, compress=False, seed=11, context=ctx)
    _, outputfiles, npack, _ = inputs.search()
    assert npack == 1000
    output = Output.restore(outputfiles[0])
    X = output.X
    GM = float(output.GM)
    r = np.sqrt(X.x.values**2 + X.y.values**2 + X.z.values**2)
    v2 = X.vx.values**2 + X.vy.values**2 + X.vz.values**2
    with np.errstate(divide='ignore', invalid='ignore'):
        energy = 0.5*v2 + GM/r
    if run == 'constant':
        assert len(X) == 1000*output.nsteps
        for i in range(0, 1000, 37):
            e = energy[(X.Index.values == i) & np.isfinite(energy) & (X.frac.values > 0)]
            assert np.all(np.isclose(e, e.mean(), rtol=2e-5))      # stored as float32
    else:
        assert len(X) == 1000
        v0 = output.X0.v.values.astype(float)
        e0 = 0.5*v0**2 + GM/1.0
        ok = np.isfinite(energy) & (X.frac.values > 0)
        assert ok.sum() > 100
        assert np.all(np.isclose(energy[ok], e0[ok], rtol=1e-3, atol=1e-12))


def test_output_constant_matches_oracle_and_reference_layout(ctx, coracle):
    inputs = Input(os.path.join(PKG_INPUTS, 'Ca.isotropic.flat.input'))     # BASELINE configs[0]
    out = Output(inputs, 2000, compress=False, seed=1234, context=ctx, save=False)
    assert out.nsteps == 361 and out.totalsource == 2000*361
    assert list(out.X.columns) == ['Index', 'time', 'x', 'y', 'z', 'vx', 'vy', 'vz', 'frac',
                                   'lossfrac']
    assert len(out.X) == 2000*361
    assert float(out.aplanet) == pytest.approx(0.387098*(1-0.20563**2)/(1+0.20563))
    # same packets through the C oracle
    f = H.mercury_forces('Ca', 0.0)
    X0 = out.X0[['time', 'x', 'y', 'z', 'vx', 'vy', 'vz', 'frac']].values.astype(float)
    c = coracle.integrate_const(f, X0, 30., 360, 15., nrec=361, threads=4)
    for k, name in enumerate(['time', 'x', 'y', 'z', 'vx', 'vy', 'vz', 'frac']):
        got = out.X[name].values.reshape(2000, 361)
        assert np.array_equal(got, c['traj'][k].T), name
    # lossfrac: cumulative frac lost while active (Output.py:420-421, zero-initialised)
    frac = out.X.frac.values.reshape(2000, 361)
    lf = out.X.lossfrac.values.reshape(2000, 361)
    alive_end = frac[:, -1] > 0
    assert np.allclose(lf[alive_end, -1], 1 - frac[alive_end, -1])
    # (frac may rise within a step that crosses the shadow edge: negative tableau weights on a
    # discontinuous loss rate -- the reference does the same -- so lossfrac is not monotone)
    assert out.counters['particle_steps'] == c['work']


def test_save_applies_compress_and_float32(ctx):
    inputs = Input(os.path.join(PKG_INPUTS, 'Ca.isotropic.flat.input'))
    out = Output(inputs, 500, compress=True, seed=5, context=ctx)          # save=True
    assert np.all(out.X.frac > 0) and out.X.x.dtype == np.float32
    assert out.X.Index.dtype == np.int32
    ids, files, npack, totalsource = inputs.search()
    assert npack == 500 and totalsource == 500*361 and len(ids) == 1
    back = Output.restore(out)
    assert back.X.x.dtype == np.float64 and back.X.Index.dtype == np.int64


def test_saving_an_output_made_without_save_still_stores_float32(ctx):
    """Output(..., save=False) keeps 64-bit rows in HBM; a later out.save() must catalogue what
    the reference's save() would (float32 / int32, Output.py:528-543), so that the image of the
    catalogue equals the one of an Output saved at construction -- packet counts exactly."""
    params = {'quantity': 'radiance', 'dims': '128,128'}
    images = []
    for save in (True, False):
        inputs = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
        inputs.options.endtime = type(inputs.options.endtime)(9000., 's')
        out = Output(inputs, 4000, compress=True, seed=11, context=ctx, save=save)
        if not save:
            assert len(inputs._catalogue) == 0
            store = out.resident_rows(ctx)
            assert store is not None and not store[0].narrow
            out.save()
        assert len(inputs._catalogue) == 1
        assert out.X.x.dtype == np.float32 and out.X.frac.dtype == np.float32
        assert out.X.Index.dtype == np.int32 and out.X0.x.dtype == np.float32
        images.append((inputs.produce_image(params, context=ctx), out.X))
    (a, xa), (b, xb) = images
    assert a.packet_image.sum() > 1e4 and np.array_equal(a.packet_image, b.packet_image)
    np.testing.assert_allclose(a.image, b.image, rtol=1e-12, atol=0)
    assert len(xa) == len(xb) and all(np.array_equal(xa[c].values, xb[c].values) for c in xa)


@pytest.mark.parametrize('quantity', ['radiance', 'column'])
def test_modelimage_two_stage_equals_streaming_equals_oracle(ctx, quantity):
    """inputs.run() + produce_image() (the reference's data flow through stored float32 packets)
    and the fused streaming image must agree: identical packet counts, weights to summation
    order; both against the NumPy oracle's create_image on the stored samples."""
    inputs = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
    inputs.options.endtime = type(inputs.options.endtime)(12000., 's')
    params = {'quantity': quantity, 'dims': '96,64', 'width': '6,4', 'center': '0.5,-0.25',
              'subobslongitude': '0.7', 'subobslatitude': '0.9'}
    inputs.run(3000, packs_per_it=1500, seed=100, context=ctx)             # two chunks
    two_stage = inputs.produce_image(params, context=ctx)
    streaming = ModelImage(inputs, params, npackets=3000, packs_per_it=1500, seed=100,
                           context=ctx)
    assert two_stage.totalsource == streaming.totalsource == 3000*401
    assert np.array_equal(two_stage.packet_image, streaming.packet_image)
    np.testing.assert_allclose(two_stage.image, streaming.image, rtol=1e-11)
    assert two_stage.packet_image.sum() > 1000
    # oracle on the stored samples of both chunks
    image = np.zeros((96, 64)); counts = np.zeros((96, 64))
    for out in inputs._catalogue:
        X = Output.restore(out).X
        M = O.image_rotation(0.7, 0.9)
        img, cnt, ex, ez = O.create_image(
            X.x.values, X.y.values, X.z.values, X.vy.values, X.frac.values,
            float(out.vrplanet)/out.unit_km, M, quantity,
            two_stage.g_tables(float(out.aplanet)), [96, 64], (-2.5, 3.5), (-2.25, 1.75),
            float(two_stage.Apix), matmul=False)
        image += img; counts += cnt
    assert np.array_equal(two_stage.packet_image, counts)
    np.testing.assert_allclose(two_stage.image, image*two_stage.atoms_per_packet, rtol=1e-11)
    assert np.allclose(two_stage.xaxis, ex[:-1] + (ex[1]-ex[0])/2)
    assert two_stage.atoms_per_packet == 1e23/(3000*401/12000.)


def test_input_run_with_the_device_sampler_continues_one_index_space(ctx):
    """Input.run(sampler='device'): the Outputs of a run take consecutive slices of one
    counter space, so the catalogue holds exactly the packets one big device draw would give and
    the two-stage image equals the streaming image of the same seed."""
    inputs = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
    inputs.options.endtime = type(inputs.options.endtime)(9000., 's')
    # the reference's chunk arithmetic: ceil(5000/2000) Outputs of 2000 packets (Input.py:228-230)
    inputs.run(5000, packs_per_it=2000, seed=3, context=ctx, sampler='device')
    assert [len(o) for o in inputs._catalogue] == [2000, 2000, 2000]
    with contextlib.redirect_stdout(io.StringIO()):
        whole = Output(inputs, 6000, seed=3, integrate=False, save=False, context=ctx,
                       sampler='device')
    for c in ('x', 'vy', 'time'):
        stored = np.concatenate([o.X0[c].values for o in inputs._catalogue])
        assert np.array_equal(stored, whole.X0[c].values.astype(np.float32)), c
    params = {'quantity': 'radiance', 'dims': '64,64'}
    two_stage = inputs.produce_image(params, context=ctx)
    streaming = ModelImage(inputs, params, npackets=6000, packs_per_it=2000, seed=3, context=ctx,
                           sampler='device')
    assert two_stage.totalsource == streaming.totalsource
    assert np.array_equal(two_stage.packet_image, streaming.packet_image)
    np.testing.assert_allclose(two_stage.image, streaming.image, rtol=1e-11)
    assert two_stage.packet_image.sum() > 1000


def _orbit(nspec, seed=0):
    """Synthetic spacecraft geometry: positions on an eccentric polar orbit (1.1 .. 3 R), looking
    in assorted directions (some at the planet, some at the limb, some away)."""
    rng = np.random.default_rng(seed)
    th = np.linspace(0, 2*np.pi, nspec, endpoint=False)
    r = 1.6 + 1.3*np.cos(th)**2
    pos = np.stack([0.3*r*np.cos(th), r*np.sin(th)*0.6 - 0.4, r*np.sin(th)*0.8], 1)
    look = rng.normal(size=(nspec, 3))
    look[::3] = -pos[::3] + 0.9*rng.normal(size=(len(pos[::3]), 3))       # roughly planetward
    look /= np.linalg.norm(look, axis=1)[:, None]
    return pos, look


def test_los_cones_match_oracle(ctx):
    """f-1: LOSResult through the GPU against the NumPy + KDTree restatement of
    compute_iteration.py: identical packet counts and `included` flags, radiance to 1e-10."""
    from nexoclom_amd import LOSResult, SpacecraftData
    inputs = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
    inputs.options.endtime = type(inputs.options.endtime)(9000., 's')
    inputs.run(4000, packs_per_it=2000, seed=21, context=ctx)
    pos, look = _orbit(150)
    sc = SpacecraftData(pos[:, 0], pos[:, 1], pos[:, 2], look[:, 0], look[:, 1], look[:, 2])
    dphi = np.radians(3.0)
    los = LOSResult(sc, inputs, {'quantity': 'radiance'}, dphi=dphi, context=ctx)
    los.simulate_data_from_inputs(sc)
    scd = {k: sc.data[k].values for k in sc.data.columns}
    rad = np.zeros(150); npk = np.zeros(150, dtype=np.int64)
    for out, it in zip(inputs._catalogue, los.iterations):
        X = Output.restore(out).X
        smp = dict(x=X.x.values, y=X.y.values, z=X.z.values, vy=X.vy.values, frac=X.frac.values,
                   Index=X.Index.values)
        r, n, inc, used = O.los_iteration(smp, scd, dphi, inputs.options.outeredge,
                                          float(out.vrplanet)/out.unit_km,
                                          los.g_tables(float(out.aplanet)), out.unit_km*1e5,
                                          n_index=2000)
        assert np.array_equal(it['npackets'].values, n)
        assert np.array_equal(it['included'], inc)
        np.testing.assert_allclose(it['radiance'].values, r, rtol=1e-10, atol=0)
        rad += r; npk += n
    assert npk.sum() > 2000 and (npk > 0).sum() > 50
    assert np.array_equal(los.npackets_los.values, npk)
    np.testing.assert_allclose(los.radiance.values, rad*los.atoms_per_packet/1e3, rtol=1e-10)


def test_los_used_pairs_and_tiles(ctx):
    """More spectra than one LDS tile (512), small cone, and the (spectrum, sample) pair list."""
    from nexoclom_amd import LOSResult, SpacecraftData
    inputs = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
    inputs.options.endtime = type(inputs.options.endtime)(6000., 's')
    inputs.run(3000, seed=5, context=ctx)
    pos, look = _orbit(700, seed=3)
    sc = SpacecraftData(pos[:, 0], pos[:, 1], pos[:, 2], look[:, 0], look[:, 1], look[:, 2])
    dphi = np.radians(1.0)
    los = LOSResult(sc, inputs, dphi=dphi, context=ctx)
    out = inputs._catalogue[0]
    assert out.resident_rows(ctx) is not None     # rows in HBM: nxc_los_accumulate_rows
    it = los.compute_iteration(out, sc, used_cap=200000)
    out._spill()                                  # rows on the host as saved (float32 frame):
    assert out.X.x.dtype == np.float32 and out.resident_rows(ctx) is None
    host32 = los.compute_iteration(out, sc, used_cap=200000)      # nxc_los_accumulate_f32
    assert np.array_equal(host32['npackets'].values, it['npackets'].values)
    assert np.array_equal(host32['included'], it['included']) and host32['n_used'] == it['n_used']
    np.testing.assert_allclose(host32['radiance'].values, it['radiance'].values, rtol=1e-12, atol=0)
    X = Output.restore(out).X
    assert out.X.x.dtype == np.float64            # restored: nxc_los_accumulate, same answer
    again = los.compute_iteration(out, sc, used_cap=200000)
    assert np.array_equal(again['npackets'].values, it['npackets'].values)
    assert np.array_equal(again['included'], it['included']) and again['n_used'] == it['n_used']
    np.testing.assert_allclose(again['radiance'].values, it['radiance'].values, rtol=1e-12, atol=0)
    smp = dict(x=X.x.values, y=X.y.values, z=X.z.values, vy=X.vy.values, frac=X.frac.values,
               Index=X.Index.values)
    scd = {k: sc.data[k].values for k in sc.data.columns}
    r, n, inc, used = O.los_iteration(smp, scd, dphi, 25., float(out.vrplanet)/out.unit_km,
                                      los.g_tables(float(out.aplanet)), out.unit_km*1e5,
                                      n_index=3000)
    assert (n[512:] > 0).sum() > 20                # the second tile of spectra sees packets too
    assert np.array_equal(it['npackets'].values, n)
    np.testing.assert_allclose(it['radiance'].values, r, rtol=1e-10, atol=0)
    pairs = set(zip(it['used'][0].tolist(), it['used'][1].tolist()))
    ref_pairs = {(i, int(row)) for i, rows in enumerate(used) for row in rows}
    assert it['n_used'] == len(ref_pairs) and pairs == ref_pairs


@pytest.mark.parametrize('layout', ['packets', 'no-index', 'scattered-index', 'tiny-packets'])
def test_los_block_formation_follows_any_index_column(ctx, layout, monkeypatch):
    """k_los cuts the samples into blocks of at most 8 rows of ONE packet (the index column says
    where packets end) and culls on the blocks' and groups' bounding spheres.  Whatever the
    column looks like -- packets of 1..40 rows, packets of 1..3 rows, no column at all, ids in no
    order (every row its own block) -- the pairs that are decided must be the reference's: counts,
    `included` and the pair list exactly, radiance to 1e-10, against the KDTree restatement.  The
    sample count is no multiple of anything, so ranges, blocks and groups end ragged."""
    from nexoclom_amd.LOSResult import POSITION, BORESIGHT, arccos_threshold, los_geometry
    import pandas as pd
    rng = np.random.default_rng({'packets': 1, 'no-index': 2, 'scattered-index': 3,
                                 'tiny-packets': 4}[layout])
    f = H.mercury_forces('Na', 1.3)
    top = 3 if layout == 'tiny-packets' else 40
    lens = rng.integers(1, top + 1, 6000 if layout == 'tiny-packets' else 2500)
    P = int(lens.sum()) - 3
    ids = np.repeat(np.arange(len(lens)), lens)[:P]
    # every packet a short straight flight from a random point near the planet
    start = rng.normal(0, 1.5, (len(lens), 3))
    vel = rng.normal(0, 0.04, (len(lens), 3))
    k = np.arange(len(ids) + 3)[:P] - np.repeat(np.cumsum(lens) - lens, lens)[:P]
    pts = start[ids] + vel[ids]*k[:, None]
    vy = rng.normal(0, 2e-3, P)
    frac = rng.uniform(0.05, 1, P)
    if layout == 'scattered-index':
        order = rng.permutation(P)
        pts, vy, frac, ids = pts[order], vy[order], frac[order], ids[order]
    pos, look = _orbit(200, seed=9)
    dphi = np.radians(2.0)
    spectra = pd.DataFrame(dict(zip(POSITION + BORESIGHT, list(pos.T) + list(look.T))))
    cut, lengths, ladder = los_geometry(spectra, 25., dphi)
    sc = np.vstack([pos.T, look.T, cut, lengths.astype(float)])
    # ('packets': four emission lines -- 66 KB of g-value tables, which leave k_los a smaller tile
    # of spectra in LDS: 200 spectra then take two tiles)
    gt = H.g_tables('Na', f.aplanet, f.R_km,
                    (3303, 5891, 5897, 5891) if layout == 'packets' else (5891, 5897))
    setup = (dphi, np.sin(dphi), np.sin(2*dphi), arccos_threshold(dphi), f.vrplanet, f.R_km*1e5,
             gt, ladder, sc)
    cols = [np.ascontiguousarray(c) for c in (pts[:, 0], pts[:, 1], pts[:, 2], vy, frac)]
    index = None if layout == 'no-index' else ids.astype(np.int64)
    n_index = P if index is None else len(lens)
    res = ctx.los_accumulate(*setup, *cols, index=index, n_index=n_index, used_cap=400000)
    tests = ctx.counters()['samples']
    # the samples go through in slabs of 2^24 rows; forced to three slabs here, the answers -- row
    # numbers of the pair list and `included` slots too -- must not change
    monkeypatch.setenv('NXC_TEST_LOS_SLAB_ROWS', str(P//3 + 1))
    slabs = ctx.los_accumulate(*setup, *cols, index=index, n_index=n_index, used_cap=400000)
    monkeypatch.delenv('NXC_TEST_LOS_SLAB_ROWS')
    assert np.array_equal(slabs['npackets'], res['npackets']) and slabs['n_used'] == res['n_used']
    assert np.array_equal(slabs['included'], res['included'])
    assert set(zip(*slabs['used'].tolist())) == set(zip(*res['used'].tolist()))
    np.testing.assert_allclose(slabs['radiance'], res['radiance'], rtol=1e-12, atol=0)
    smp = dict(x=cols[0], y=cols[1], z=cols[2], vy=vy, frac=frac,
               Index=np.arange(P) if index is None else index)
    scd = {c: spectra[c].values for c in spectra.columns}
    r, n, inc, used = O.los_iteration(smp, scd, dphi, 25., f.vrplanet, gt, f.R_km*1e5,
                                      n_index=n_index)
    assert n.sum() > 300
    assert np.array_equal(res['npackets'], n) and np.array_equal(res['included'], inc)
    np.testing.assert_allclose(res['radiance'], r, rtol=1e-10, atol=0)
    pairs = set(zip(res['used'][0].tolist(), res['used'][1].tolist()))
    assert res['n_used'] == len(pairs) == sum(len(u) for u in used)
    assert pairs == {(i, int(row)) for i, rows in enumerate(used) for row in rows}
    # the culling culls: fewer sphere tests than (block, spectrum) pairs even for this compact
    # cloud seen from close by (the bench cloud: a quarter) -- except where every row is a block
    # of its own
    if layout == 'packets':
        assert tests < 0.8*((P + 7)//8)*200, tests
    # a non-finite row goes to the exact test (which drops it like the reference), its neighbours
    # are decided as before
    cols[0][1000] = np.nan
    cols[2][2000] = np.inf
    res2 = ctx.los_accumulate(*setup, *cols, index=index, n_index=n_index)
    ok = np.ones(P, bool); ok[[1000, 2000]] = False
    smp2 = {k_: v[ok] for k_, v in smp.items()}
    r2, n2, inc2, _ = O.los_iteration(smp2, scd, dphi, 25., f.vrplanet, gt, f.R_km*1e5,
                                      n_index=n_index)
    assert np.array_equal(res2['npackets'], n2)
    np.testing.assert_allclose(res2['radiance'], r2, rtol=1e-10, atol=0)


def test_los_with_more_spectra_than_the_lds_sums_hold(ctx):
    """2800 lines of sight: six LDS tiles of spectra in k_los, and too many for k_los_pairs' per-
    workgroup sums in LDS (it then adds straight to memory).  Same answers as the restatement."""
    from nexoclom_amd.LOSResult import POSITION, BORESIGHT, arccos_threshold, los_geometry
    import pandas as pd
    rng = np.random.default_rng(12)
    f = H.mercury_forces('Na', 1.3)
    lens = rng.integers(5, 60, 600)
    P = int(lens.sum())
    ids = np.repeat(np.arange(len(lens)), lens)
    start = rng.normal(0, 1.8, (len(lens), 3))
    vel = rng.normal(0, 0.03, (len(lens), 3))
    k = np.arange(P) - np.repeat(np.cumsum(lens) - lens, lens)
    pts = start[ids] + vel[ids]*k[:, None]
    vy, frac = rng.normal(0, 2e-3, P), rng.uniform(0.05, 1, P)
    S = 2800
    pos, look = _orbit(S, seed=4)
    dphi = np.radians(1.5)
    spectra = pd.DataFrame(dict(zip(POSITION + BORESIGHT, list(pos.T) + list(look.T))))
    cut, lengths, ladder = los_geometry(spectra, 25., dphi)
    sc = np.vstack([pos.T, look.T, cut, lengths.astype(float)])
    gt = H.g_tables('Na', f.aplanet, f.R_km, (5891, 5897))
    cols = [np.ascontiguousarray(c) for c in (pts[:, 0], pts[:, 1], pts[:, 2], vy, frac)]
    res = ctx.los_accumulate(dphi, np.sin(dphi), np.sin(2*dphi), arccos_threshold(dphi), f.vrplanet,
                             f.R_km*1e5, gt, ladder, sc, *cols, index=ids.astype(np.int64),
                             n_index=len(lens))
    smp = dict(x=cols[0], y=cols[1], z=cols[2], vy=vy, frac=frac, Index=ids)
    scd = {c: spectra[c].values for c in spectra.columns}
    r, n, inc, _ = O.los_iteration(smp, scd, dphi, 25., f.vrplanet, gt, f.R_km*1e5, n_index=len(lens))
    assert n.sum() > 1000 and (n[2560:] > 0).sum() > 10          # the last tile sees packets too
    assert np.array_equal(res['npackets'], n) and np.array_equal(res['included'], inc)
    np.testing.assert_allclose(res['radiance'], r, rtol=1e-10, atol=0)


def test_device_sampler_matches_philox_oracle_and_reference_statistics(ctx):
    """f-4: k_sample == NumPy Philox restatement (to libm rounding), is counter-addressed
    (chunks concatenate), and is statistically the reference's source (KS tests in the spirit of
    tests/unit_tests/Initial_state/test_spatial_distribution.py:95-143)."""
    from scipy import stats
    inputs = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
    out = Output(inputs, 200000, seed=77, integrate=False, save=False, context=ctx,
                 sampler='device')
    X = out.X0[['time', 'x', 'y', 'z', 'vx', 'vy', 'vz', 'frac']].values
    ref = O.sample_x0_philox(200000, 77, **out.source_desc())
    np.testing.assert_allclose(X, ref, rtol=1e-12, atol=1e-15)
    a = ctx.sample_packets(1000, 77, first_index=0, download=True, **out.source_desc())
    b = ctx.sample_packets(1000, 77, first_index=1000, download=True, **out.source_desc())
    c = ctx.sample_packets(2000, 77, first_index=0, download=True, **out.source_desc())
    assert np.array_equal(np.concatenate([a, b], axis=1), c)
    # against the host sampler with NumPy's generator: same distributions
    host = Output(inputs, 200000, seed=5, integrate=False, save=False)
    for col in ('x', 'y', 'z', 'vx', 'vy', 'vz'):
        assert stats.ks_2samp(out.X0[col].values, host.X0[col].values).pvalue > 1e-3, col
    lon = (np.arctan2(X[:, 1], -X[:, 2]) + 2*np.pi) % (2*np.pi)
    assert stats.kstest(lon, 'uniform', args=(0, 2*np.pi)).pvalue > 1e-3
    assert stats.kstest(X[:, 3], 'uniform', args=(-1, 2)).pvalue > 1e-3          # sin(lat)
    speed = np.linalg.norm(X[:, 4:7], axis=1)*out.unit_km
    assert stats.kstest(speed, 'uniform', args=(0.5, 4.0)).pvalue > 1e-3
    sinalt = np.sum(X[:, 1:4]*X[:, 4:7], axis=1)/np.linalg.norm(X[:, 4:7], axis=1)
    assert stats.kstest(sinalt, 'uniform', args=(0, 1)).pvalue > 1e-3


def test_device_sampled_image_matches_oracle(ctx, coracle):
    """Streaming image from device-sampled packets == C oracle on the same (downloaded) X0."""
    inputs = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
    params = {'quantity': 'radiance', 'dims': '64,64'}
    img = ModelImage(inputs, params, npackets=6000, packs_per_it=3000, seed=9, context=ctx,
                     sampler='device')
    out = Output(inputs, 6000, seed=9, integrate=False, save=False, context=ctx, sampler='device')
    X0 = out.X0[['time', 'x', 'y', 'z', 'vx', 'vy', 'vz', 'frac']].values
    f = H.mercury_forces('Na', 1.3)
    im = H.image_setup(f, 'radiance', dims=(64, 64))
    desc = coracle.image_desc(im['M'], f.vrplanet, im['apix'], 'radiance', im['g_tables'],
                              im['xedges'], im['zedges'], downcast=True)
    ref = coracle.integrate_const(f, X0, 30., 1667, 25., img=desc, threads=4)
    assert np.array_equal(img.packet_image, ref['counts'].astype(float))
    np.testing.assert_allclose(img.image, ref['image']*img.atoms_per_packet, rtol=1e-11)


@pytest.mark.parametrize('infile', ['Bounce.const.input', 'Bounce.tempdep.input'])
def test_surface_reemission_matches_oracle(ctx, infile):
    """f-2: packets that hit the surface are re-emitted (bouncepackets.py).  GPU vs the NumPy
    restatement driven by the same Philox uniforms: trajectories to 1e-8, identical bounce
    bookkeeping; plus the physics the reference intends (packets stay above the surface, frac
    only drops by the sticking factor at an impact)."""
    from nexoclom_amd.surface import bounce_config
    inputs = Input(os.path.join(HERE, 'inputfiles', infile))
    n = 400
    out = Output(inputs, n, compress=False, seed=31, context=ctx, save=False)
    nsteps = out.nsteps
    traj = np.stack([out.X[c].values.reshape(n, nsteps) for c in
                     ['time', 'x', 'y', 'z', 'vx', 'vy', 'vz', 'frac']], axis=1)   # (n, 8, nsteps)
    f = H.mercury_forces('Na', float(inputs.geometry.taa), True, inputs.forces.radpres, 0.0)
    X0 = out.X0[['time', 'x', 'y', 'z', 'vx', 'vy', 'vz', 'frac']].values.astype(float)
    cfg = bounce_config(inputs, f.GM, f.R_km, 31)
    res, nb, work = O.constant_step_driver_bounce(f, X0, inputs.options.endtime.value, 30.,
                                                  inputs.options.outeredge, cfg)
    assert nb.sum() > n                     # most packets come back down at least once
    assert out.counters['particle_steps'] == work
    np.testing.assert_allclose(traj, res, rtol=1e-8, atol=1e-10)
    r = np.sqrt(traj[:, 1]**2 + traj[:, 2]**2 + traj[:, 3]**2)
    alive = traj[:, 7] > 0
    assert np.all(r[alive] > 1 - 1e-9)      # never left inside the planet
    # the fused (lane-refill) kernel re-emits identically: same final states as the trajectory run
    ctx.upload_packets(X0)
    ctx.set_bounce(cfg)
    ctx.set_first_index(0)
    g = ctx.integrate_const(30., nsteps - 1, inputs.options.outeredge, want_final=True,
                            want_steps=True)
    last = np.minimum(g['steps'], nsteps - 1)
    assert np.array_equal(g['final'], traj[np.arange(n), :, last])
    ctx.set_bounce(None)


def test_reemission_statistics_follow_the_reference_process(ctx):
    """Rebound directions are cosine-weighted in altitude and uniform in azimuth
    (bouncepackets.py:8-18), speeds follow the accommodation law (:77-78): checked on the first
    bounce of many packets dropped onto the day side."""
    from scipy import stats
    from nexoclom_amd.surface import bounce_config
    inputs = Input(os.path.join(HERE, 'inputfiles', 'Bounce.const.input'))
    inputs.surfaceinteraction.accomfactor = 0.0          # elastic: |v| after = impact speed
    n = 40000
    rng = np.random.default_rng(4)
    X0 = np.zeros((n, 8)); X0[:, 0] = 9000.; X0[:, 7] = 1.0
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    X0[:, 1:4] = d*1.001                                  # just above the surface, moving down
    X0[:, 4:7] = -d*1.5/2440.53
    out = Output(inputs, n, seed=1, integrate=False, save=False, context=ctx)
    cfg = bounce_config(inputs, float(out.GM), out.unit_km, 123)
    ctx.set_forces(**out.forces_kwargs())
    ctx.upload_packets(X0); ctx.set_bounce(cfg); ctx.set_first_index(0)
    g = ctx.integrate_const(30., 1, 15., nrec=2)
    ctx.set_bounce(None)
    x1, v1, f1 = g['traj'][1:4, 1].T, g['traj'][4:7, 1].T, g['traj'][7, 1]
    assert np.allclose(np.linalg.norm(x1, axis=1), 1.0, atol=1e-12)      # put back on the surface
    assert np.all((f1 <= 0.7 + 1e-12) & (f1 > 0.69))    # (1 - stickcoef) x one step of photo-loss
    speed = np.linalg.norm(v1, axis=1)
    sinalt = np.sum(x1*v1, axis=1)/speed
    assert stats.kstest(sinalt, 'uniform', args=(0, 1)).pvalue > 1e-3
    east = np.stack([x1[:, 1], -x1[:, 0], np.zeros(n)], 1)
    east /= np.linalg.norm(east, axis=1)[:, None]
    north = np.cross(x1, east)
    az = (np.arctan2(np.sum(v1*east, axis=1), np.sum(v1*north, axis=1)) + 2*np.pi) % (2*np.pi)
    assert stats.kstest(az, 'uniform', args=(0, 2*np.pi)).pvalue > 1e-3
    # elastic rebound: kinetic energy at the surface = impact energy (bouncepackets.py:59-66)
    GM = float(out.GM)
    r_in = np.linalg.norm(X0[:, 1:4] + 0, axis=1)
    e_in = 0.5*np.sum(X0[:, 4:7]**2, axis=1) + GM/r_in
    e_out = 0.5*speed**2 + GM/1.0
    assert np.allclose(e_out, e_in, rtol=2e-3)           # one RK step of free fall in between


def test_output_files_round_trip(ctx, tmp_path):
    """f-3: the on-disk Output (columnar float32 .npz instead of the reference's pickle): what
    restore() gives back is what save() stored (compress filter + 32-bit down-cast,
    Output.py:522-570), and ModelImage built from the files equals the in-memory one."""
    inputs = Input(os.path.join(PKG_INPUTS, 'Ca.isotropic.flat.input'), savepath=str(tmp_path))
    inputs.run(1200, packs_per_it=600, seed=3, context=ctx)
    ids, files, npack, total = inputs.search()
    assert len(files) == 2 and all(os.path.exists(f) for f in files) and npack == 1200
    params = {'quantity': 'column', 'dims': '48,48', 'width': '6,6'}
    mem = inputs.produce_image(params, context=ctx)
    image = np.zeros((48, 48)); counts = np.zeros((48, 48))
    for f, out in zip(files, inputs._catalogue):
        back = Output.restore(f)
        assert back.X.x.dtype == np.float64 and len(back.X) == len(out.X)
        assert np.array_equal(back.X.x.values, out.X.x.values.astype(np.float64))
        assert back.totalsource == out.totalsource and back.nsteps == out.nsteps
        im, ct = mem.create_image(f)
        image += im.histogram; counts += ct.histogram
    assert np.array_equal(counts, mem.packet_image)
    np.testing.assert_allclose(image*mem.atoms_per_packet, mem.image, rtol=1e-12)


@pytest.mark.parametrize('sampler', ['device', 'numpy'])
def test_sharded_image_is_independent_of_the_shard_count(ctx, sampler):
    """Multi-GPU semantics on one GPU: the image of N packets equals the sum of the images of its
    index shards (what nexoclom_amd.distributed.sharded_image reduces over RCCL), packet counts
    exactly -- i.e. 1, 2 or 8 GPUs give the same image -- for the counter-based device sampler
    and for the host sampler (fixed chunk grid, chunk k drawn from seed + k, rows sliced)."""
    from nexoclom_amd.distributed import ControlPlane, shard_range, sharded_image
    inputs = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
    params = {'quantity': 'radiance', 'dims': '128,128'}
    whole = sharded_image(inputs, params, 9001, seed=42, cp=ControlPlane(world=1, rank=0),
                          context=ctx, sampler=sampler, packs_per_it=4000)
    image = np.zeros((128, 128)); counts = np.zeros((128, 128)); total = 0.
    for rank in range(3):
        lo, hi = shard_range(9001, rank, 3)
        part = ModelImage(inputs, params, npackets=9001, shard=(lo, hi), seed=42, context=ctx,
                          sampler=sampler, packs_per_it=4000, finalize=False)
        assert part.npackets == hi - lo
        image += part.image; counts += part.packet_image; total += part.totalsource
    assert total == whole.totalsource == 9001*1668
    assert counts.sum() > 1e5
    assert np.array_equal(counts, whole.packet_image)
    np.testing.assert_allclose(image*whole.atoms_per_packet, whole.image, rtol=1e-11)
    # and the un-sharded streaming ModelImage is that same image
    plain = ModelImage(inputs, params, npackets=9001, seed=42, context=ctx, sampler=sampler,
                       packs_per_it=4000)
    assert np.array_equal(plain.packet_image, whole.packet_image)


def test_rccl_refuses_two_ranks_on_one_device(ctx):
    """Two communicator ranks on one GPU is what a mis-launched job looks like; the control plane
    reports it before RCCL is asked (and names the device), instead of timing a fallback."""
    from nexoclom_amd import hip_api
    from nexoclom_amd.distributed import ControlPlane

    class TwoOnOne(ControlPlane):          # a 2-rank world whose ranks share this process's GPU
        def __init__(self):
            super().__init__(world=1, rank=0)
            self.world = 2

        def allgather_bytes(self, payload):
            return [payload, payload]
    with pytest.raises(hip_api.HipError, match='one process per GPU'):
        TwoOnOne().init_rccl(ctx)
    assert len(ctx.bus_id()) >= 7


REF_INPUTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'inputfiles')


@pytest.mark.parametrize('infile', ['Na.reference.input', 'Ca.reference.input'])
def test_device_sampler_covers_the_reference_run_sources(ctx, infile):
    """The sources of the reference's own reference runs (tests/test_data/inputfiles/
    {Na,Ca}.reference.input: 'surface spot' + 'maxwellian', which its system test
    tests/system_tests/test_run_through.py:9-31 drives) on the device sampler: equal to the NumPy
    Philox restatement to libm rounding, counter-addressed, and statistically the host sampler's
    distributions (two-sample KS per column; the host one draws from NumPy's global generator)."""
    import numpy.random as nprandom
    from scipy import stats
    inputs = Input(os.path.join(REF_INPUTS, infile))
    n = 100000
    out = Output(inputs, n, seed=31, integrate=False, save=False, context=ctx, sampler='device')
    src = out.source_desc()
    assert src['spatial_type'] == 1 and src['speed_type'] == 2
    assert src['surface_map'].shape == (361, 181) and len(src['speed_table'][0]) == 5000
    X = out.X0[['time', 'x', 'y', 'z', 'vx', 'vy', 'vz', 'frac']].values
    ref = O.sample_x0_philox(n, 31, **src)
    np.testing.assert_allclose(X, ref, rtol=1e-11, atol=1e-14)
    a = ctx.sample_packets(700, 31, first_index=0, download=True, **src)
    b = ctx.sample_packets(300, 31, first_index=700, download=True, **src)
    c = ctx.sample_packets(1000, 31, first_index=0, download=True, **src)
    assert np.array_equal(np.concatenate([a, b], axis=1), c)
    nprandom.seed(8)
    host = Output(inputs, n, seed=5, integrate=False, save=False)
    for col in ('x', 'y', 'z', 'vx', 'vy', 'vz'):
        assert stats.ks_2samp(out.X0[col].values, host.X0[col].values).pvalue > 1e-3, col
    speed_d = np.linalg.norm(X[:, 4:7], axis=1)
    assert stats.ks_2samp(speed_d, host.X0.v.values).pvalue > 1e-3
    # the spot: launch points cluster around the requested centre
    centre = np.array([np.sin(float(inputs.spatialdist.longitude)),
                       -np.cos(float(inputs.spatialdist.longitude)), 0.0])
    assert (X[:, 1:4] @ centre).mean() > 0.3


@pytest.mark.parametrize('infile', ['Na.reference.input', 'Ca.reference.input'])
def test_reference_run_inputs_end_to_end(ctx, coracle, infile):
    """{Na,Ca}.reference.input through Input -> ModelImage(npackets=, sampler='device'): the
    streamed radiance image equals the C oracle run on the very packets the device drew (counts
    exactly, brightness to 1e-11)."""
    inputs = Input(os.path.join(REF_INPUTS, infile))
    species = inputs.options.species
    params = {'quantity': 'radiance', 'dims': '96,96', 'width': '12,12'}
    img = ModelImage(inputs, params, npackets=5000, packs_per_it=2000, seed=4, context=ctx,
                     sampler='device')
    out = Output(inputs, 5000, seed=4, integrate=False, save=False, context=ctx, sampler='device')
    X0 = out.X0[['time', 'x', 'y', 'z', 'vx', 'vy', 'vz', 'frac']].values
    opt = inputs.options
    f = H.mercury_forces(species, float(inputs.geometry.taa))
    waves = {'Na': (5891, 5897), 'Ca': (4227,)}[species]
    im = H.image_setup(f, 'radiance', dims=(96, 96), width=(12., 12.), species=species,
                       wavelengths=waves)
    desc = coracle.image_desc(im['M'], f.vrplanet, im['apix'], 'radiance', im['g_tables'],
                              im['xedges'], im['zedges'], downcast=True)
    nsteps, n_iter = O.n_output_steps(opt.endtime.value, opt.step_size)
    ref = coracle.integrate_const(f, X0, opt.step_size, n_iter, opt.outeredge, img=desc, threads=4)
    assert img.counters['particle_steps'] == ref['work']
    assert ref['counts'].sum() > 10000
    assert np.array_equal(img.packet_image, ref['counts'].astype(float))
    np.testing.assert_allclose(img.image, ref['image']*img.atoms_per_packet, rtol=1e-11)


def test_device_sampler_refuses_unusable_tables(ctx):
    """nxc_packets_sample fails loudly (NXC_ERR_ARG with a reason) instead of sampling from a
    density map that is zero everywhere or a cumulative table that decreases; a map that is zero
    almost everywhere still samples (the rejection loop is bounded and reports packets that never
    found a launch point)."""
    from nexoclom_amd import hip_api
    inputs = Input(os.path.join(REF_INPUTS, 'Na.reference.input'))
    out = Output(inputs, 10, seed=1, integrate=False, save=False)
    src = out.source_desc()
    bad = dict(src, surface_map=np.zeros((361, 181)))
    with pytest.raises(hip_api.HipError, match='all zero'):
        ctx.sample_packets(100, 1, **bad)
    cdf, v = src['speed_table']
    with pytest.raises(hip_api.HipError, match='non-decreasing'):
        ctx.sample_packets(100, 1, **dict(src, speed_table=(cdf[::-1].copy(), v)))
    with pytest.raises(hip_api.HipError, match='finite'):
        ctx.sample_packets(100, 1, **dict(src, surface_map=np.full((361, 181), np.nan)))
    spike = np.zeros((361, 181)); spike[100, 90] = 1.0          # one node of 65 341
    with pytest.raises(hip_api.HipError, match='no launch point'):
        ctx.sample_packets(20000, 1, **dict(src, surface_map=spike))
    # the failed call left no packets behind (its never-accepted candidates are not a resident set)
    with pytest.raises(hip_api.HipError, match='no resident packets'):
        ctx.integrate_const(30., 10, 25.)
    # a usable call afterwards still works on the same handle
    X = ctx.sample_packets(1000, 1, download=True, **src)
    assert np.isfinite(X).all() and np.allclose(np.linalg.norm(X[1:4], axis=0), 1.0)
    # a narrow spot (sigma 0.05 rad: the uniform proposal is accepted 8 times in 10 000) needs far
    # more than a fixed few thousand trials for the unluckiest of many packets: the budget follows
    # the map's acceptance rate, and the packets land around the spot
    from nexoclom_amd.source_distribution import spot_density_map
    lon0, lat0 = 1.0, 0.3
    _, _, narrow = spot_density_map(lon0, lat0, 0.05)
    X = ctx.sample_packets(300000, 5, download=True, **dict(src, surface_map=narrow))
    # (the reference's map is built with -sin(lat) for z, source_distribution.py:96-113: the spot
    # sits at latitude -lat0)
    centre = np.array([np.sin(lon0)*np.cos(lat0), -np.cos(lon0)*np.cos(lat0), -np.sin(lat0)])
    far = np.arccos(np.clip(centre @ X[1:4], -1, 1))
    assert np.isfinite(X).all() and np.median(far) < 0.12 and (far < 0.5).mean() > 0.99


@pytest.mark.parametrize('mode', ['constant', 'variable', 'device-sampled', 'device-variable'])
def test_batched_input_run_equals_output_by_output(ctx, mode, tmp_path):
    """Input.run integrates the Outputs of a pass in ONE launch (Output.integrate_batch) and
    leaves their rows in HBM: every Output must hold exactly what it holds when the Outputs are
    integrated one by one (batch=False) -- frames, labels, dtypes, totals -- the image built from
    the resident rows must be the image built from the host frames, and the files written by the
    worker thread must restore to the same frames."""
    def make(batch, savepath=None):
        inputs = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'), savepath=savepath)
        inputs.options.endtime = type(inputs.options.endtime)(9000., 's')
        if 'variable' in mode:
            inputs.options.step_size = 0.
            inputs.options.resolution = 1e-4
        with contextlib.redirect_stdout(io.StringIO()):
            inputs.run(4200, packs_per_it=1000, seed=9, context=ctx, batch=batch,
                       sampler='device' if mode.startswith('device') else 'numpy')
        return inputs
    one, many = make(False), make(True, str(tmp_path))
    assert len(one._catalogue) == len(many._catalogue) == 5
    for a, b in zip(one._catalogue, many._catalogue):
        if 'variable' not in mode:
            assert b.resident_rows(ctx) is not None and b._X is None       # nothing on the host yet
        assert a.totalsource == b.totalsource and a.nsteps == b.nsteps and a.idnum == b.idnum
        assert list(a.X.columns) == list(b.X.columns)
        assert np.array_equal(a.X.index.values, b.X.index.values)
        for c in a.X.columns:
            assert a.X[c].dtype == b.X[c].dtype, c
            assert np.array_equal(a.X[c].values, b.X[c].values), c
        for c in a.X0.columns:
            assert np.array_equal(a.X0[c].values, b.X0[c].values), c
        back = Output.restore(b.filename)
        for c in ('x', 'vy', 'frac', 'Index'):
            assert np.array_equal(back.X[c].values, b.X[c].values.astype(back.X[c].dtype)), c
    if 'variable' not in mode:
        params = {'quantity': 'radiance', 'dims': '64,64'}
        resident = many.produce_image(params, context=ctx)       # rows read in HBM
        for b in many._catalogue:
            b._spill()                                           # now from the host frames
            assert b.resident_rows(ctx) is None
        host = many.produce_image(params, context=ctx)
        assert resident.packet_image.sum() > 1000
        assert np.array_equal(resident.packet_image, host.packet_image)
        np.testing.assert_allclose(resident.image, host.image, rtol=1e-12, atol=0)
        one_img = one.produce_image(params, context=ctx)
        assert np.array_equal(one_img.packet_image, host.packet_image)


def test_device_pcg64_follows_the_seeded_host_stream(ctx):
    """generator='pcg64' (nxc_source_desc.generator = 1): the device jumps into NumPy's PCG64
    stream (Output.py:92; the reference draws whole random(npackets) vectors one after the other,
    source_distribution.py:51-62,169-171,202-212) instead of using its own Philox counters.
    The uniforms are bit-equal to default_rng(seed).random(n), window by window; X0 equals the
    host sampler's to libm rounding; the packet-count image of a 2e4-packet run is the
    host-sampled run's; shards and Input.run follow."""
    # 1. the raw uniforms: six successive vectors, whole and windows, small and bench-sized n
    for seed, n, row0, count in ((1234, 1000, 0, 1000), (1234, 1000, 123, 754), (7, 1, 0, 1),
                                 (99, 10_000_000, 0, 4096), (99, 10_000_000, 9_990_000, 10_000),
                                 (2**63 + 5, 20_000_000, 12_345_678, 3000)):
        rng = np.random.default_rng(seed)
        got = ctx.pcg64_uniforms(seed, n, row0, count, 6)
        for v in range(6):
            assert np.array_equal(got[v], rng.random(n)[row0:row0 + count]), (seed, n, v)
    # 2. X0: constant-step (five vectors) and variable-step (launch times first) sources
    eps = np.finfo(float).eps
    for variable in (False, True):
        inputs = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
        if variable:
            inputs.options.step_size = 0.
            inputs.options.resolution = 1e-4
        n, seed = 20000, 321
        with contextlib.redirect_stdout(io.StringIO()):
            host = Output(inputs, n, seed=seed, integrate=False, save=False, context=ctx)
            dev = Output(inputs, n, seed=seed, integrate=False, save=False, context=ctx,
                         sampler='device', generator='pcg64')
            part = Output(inputs, n, seed=seed, integrate=False, save=False, context=ctx,
                          sampler='device', generator='pcg64', window=(n, 777, 15001))
        speed = host.X0.v.values.max()
        for c in ('time', 'x', 'y', 'z', 'vx', 'vy', 'vz', 'frac'):
            a, b = dev.X0[c].values, host.X0[c].values
            scale = speed if c.startswith('v') else (1.0 if c in 'xyz' else np.abs(b).max())
            assert np.abs(a - b).max() <= 4*eps*scale, (c, np.abs(a - b).max()/eps/scale)
            assert np.array_equal(part.X0[c].values, a[777:15001]), c      # a window = a slice
        assert (dev.X0.x.values == host.X0.x.values).mean() > 0.5           # mostly the same bits
        if variable:
            assert np.array_equal(dev.X0.time.values, host.X0.time.values)  # u * endtime: no libm
    # 3. the run: host-sampled vs device-sampled packets through the streaming image
    inputs = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
    params = {'quantity': 'radiance', 'dims': '128,128'}
    kw = dict(npackets=20000, seed=55, packs_per_it=8000, context=ctx)
    with contextlib.redirect_stdout(io.StringIO()):
        host_img = ModelImage(inputs, params, sampler='numpy', **kw)
        dev_img = ModelImage(inputs, params, sampler='device', generator='pcg64', **kw)
        shards = [ModelImage(inputs, params, sampler='device', generator='pcg64', finalize=False,
                             shard=s, **kw) for s in ((0, 6001), (6001, 14444), (14444, 20000))]
    assert host_img.packet_image.sum() > 1e6
    assert np.array_equal(dev_img.packet_image, host_img.packet_image)
    np.testing.assert_allclose(dev_img.image, host_img.image, rtol=1e-9, atol=0)
    assert dev_img.counters['particle_steps'] == host_img.counters['particle_steps']
    assert np.array_equal(sum(s.packet_image for s in shards), host_img.packet_image)
    # 4. Input.run: Output k follows default_rng(seed + k), all Outputs in one launch
    runs = {}
    for sampler in ('numpy', 'device'):
        inp = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
        inp.options.endtime = type(inp.options.endtime)(6000., 's')
        with contextlib.redirect_stdout(io.StringIO()):
            inp.run(5000, packs_per_it=2000, seed=8, context=ctx, sampler=sampler,
                    **({'generator': 'pcg64'} if sampler == 'device' else {}))
        runs[sampler] = inp
    for a, b in zip(runs['numpy']._catalogue, runs['device']._catalogue):
        np.testing.assert_allclose(b.X0.x.values, a.X0.x.values, rtol=0, atol=1e-6)   # float32 now
        assert len(a.X) == len(b.X) and np.array_equal(a.X.Index.values, b.X.Index.values)
        np.testing.assert_allclose(b.X.x.values, a.X.x.values, rtol=0, atol=1e-5)


def test_los_column_through_a_uniform_shell_is_the_chord_length(ctx):
    """Physics pin of f-1 (parity unpinned: compute_iteration.py needs astropy / PostgreSQL /
    MESSENGERuvvs and the reference holds no line-of-sight fixture).  Samples of equal weight
    distributed uniformly in the shell 1.2 R < r < 3 R have a closed-form answer for every line of
    sight: a cone of half-angle dphi collects w / Apix(d) from each sample at distance d, Apix =
    pi (d sin dphi)^2 (compute_iteration.py:194-196), so

        radiance = n w / R_cm^2 * 2 / (1 + cos dphi) * L         n = samples per R^3,

    L = length of the line inside the shell (up to the spacecraft-planet distance if it hits the
    planet, compute_iteration.py:105-115,185), and the number of samples seen is
    n * 2 pi (1 - cos dphi) * (d_out^3 - d_in^3) / 3 per crossing.  Checked to Monte-Carlo error
    through LOSResult for lines that cross the shell once, twice, graze it and miss it."""
    from nexoclom_amd import LOSResult, SpacecraftData
    rng = np.random.default_rng(2026)
    r1, r2, N = 1.2, 3.0, 6_000_000
    r = np.cbrt(rng.uniform(r1**3, r2**3, N))
    u = rng.normal(size=(3, N))
    xyz = (u/np.linalg.norm(u, axis=0)*r).astype(np.float32)
    density = N/(4/3*np.pi*(r2**3 - r1**3))
    inputs = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
    out = Output.__new__(Output)                       # a stored Output made of these samples
    out.inputs, out.npackets, out.totalsource = inputs, N, float(N)
    out.idnum, out.filename = 1, None
    import pandas as pd
    from nexoclom_amd.units import Quantity
    out.X = pd.DataFrame({'Index': np.arange(N, dtype=np.int32), 'x': xyz[0], 'y': xyz[1],
                          'z': xyz[2], 'vy': np.zeros(N, np.float32),
                          'frac': np.ones(N, np.float32)})
    out.X0 = pd.DataFrame({'x': np.zeros(N, np.float32)})
    out.aplanet, out.vrplanet = Quantity(0.3514, 'au'), Quantity(0.0, 'km/s')
    out.unit_km = inputs.geometry.planet.radius.value
    inputs._catalogue.append(out)
    # spacecraft 8 R sunward of the planet (y = -8: what it sees in front of the planet is lit),
    # looking along +y at impact parameters b
    D, dphi = 8.0, np.radians(1.0)
    b = np.array([0.0, 0.6, 1.1, 1.19, 2.0, 2.7, 3.3])
    sc = SpacecraftData(b, np.full_like(b, -D), np.zeros_like(b), np.zeros_like(b),
                        np.ones_like(b), np.zeros_like(b))
    g = 1.7
    los = LOSResult(sc, inputs, {'quantity': 'radiance', 'g': str(g)}, dphi=dphi, context=ctx)
    it = los.compute_iteration(out, sc)

    def crossings(bi):
        """[(d_in, d_out)] along the line, from the spacecraft."""
        if bi >= r2:
            return []
        far = np.sqrt(r2**2 - bi**2)
        if bi >= r1:
            return [(D - far, D + far)]
        near = np.sqrt(r1**2 - bi**2)
        front = [(D - far, D - near)]
        return front if bi < 1.0 else front + [(D + near, D + far)]   # the planet hides the back
    R_cm = out.unit_km*1e5
    w = 1.0*g/1e6                                      # frac * g / 1e6 (ModelResult.py:148-161)
    for k, bi in enumerate(b):
        segs = crossings(bi)
        length = sum(d1 - d0 for d0, d1 in segs)
        want = density*w/R_cm**2 * 2/(1 + np.cos(dphi)) * length
        count = density*2*np.pi*(1 - np.cos(dphi))*sum(d1**3 - d0**3 for d0, d1 in segs)/3
        got, seen = it['radiance'].values[k], it['npackets'].values[k]
        if not segs:
            assert got == 0 and seen == 0
            continue
        # Monte-Carlo error of `seen` samples (the weights w / Apix vary by < 2 along a crossing),
        # plus the variation of the chord across the cone where the line runs close to an edge
        sigma = 4/np.sqrt(count) + (0.06 if min(abs(bi - r1), abs(bi - r2), abs(bi - 1.0)) < 0.2
                                    else 0.0)
        assert abs(seen/count - 1) < sigma, (bi, seen, count)
        assert abs(got/want - 1) < sigma, (bi, got, want)
    assert it['npackets'].values[0] > 2000             # the bound above is a few per cent


def test_row_stores_spill_to_the_host_when_hbm_is_needed(ctx):
    """The catalogue's rows stay in HBM only while there is room: before a new store is built
    the oldest are spilled to their Outputs' host memory (Context.make_room), and the Outputs --
    frames, images, line-of-sight runs -- keep working from the host copy."""
    inputs = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
    inputs.options.endtime = type(inputs.options.endtime)(6000., 's')
    with contextlib.redirect_stdout(io.StringIO()):
        inputs.run(3000, packs_per_it=1000, seed=4, context=ctx)
    outs = inputs._catalogue
    assert all(o.resident_rows(ctx) is not None and o._X is None for o in outs)
    # (host-drawn Outputs are launched as they become ready: one store per launch)
    stores = list({id(o._store): o._store for o in outs}.values())
    assert sum(s.total for s in stores) == sum(o._nrows for o in outs)
    params = {'quantity': 'column', 'dims': '64,64'}
    before = inputs.produce_image(params, context=ctx)
    free, total = ctx.mem_info()
    ctx.make_room(total)                       # more than can ever be free: everything spills
    assert all(s._r is None for s in stores) and all(o.resident_rows(ctx) is None for o in outs)
    assert all(o._X is not None and len(o.X) == o._nrows for o in outs)     # rows are on the host now
    after = inputs.produce_image(params, context=ctx)
    assert np.array_equal(before.packet_image, after.packet_image) and before.packet_image.sum() > 100
    np.testing.assert_allclose(after.image, before.image, rtol=1e-12, atol=0)
    assert ctx.mem_info()[0] >= free           # the store's memory came back
    # and the next run builds a new store as if nothing had happened
    with contextlib.redirect_stdout(io.StringIO()):
        inputs.run(4000, packs_per_it=1000, seed=4, context=ctx)
    assert len(inputs._catalogue) == 4 and inputs._catalogue[-1].resident_rows(ctx) is not None


def test_blocks_of_freed_row_stores_are_reused(ctx, coracle):
    """A freed store's device blocks wait in the handle for the next store of about that size
    (nxc_rows_free); they count as free memory, are handed out again with the new rows intact,
    and go back to the driver when something else needs the memory (not forced here)."""
    f = H.mercury_forces('Na', 1.3)
    H.set_ctx_forces(ctx, f)
    endtime, step = 9000., 30.
    nsteps, n_iter = O.n_output_steps(endtime, step)
    X0 = H.sample_x0(400_000, 5, endtime)            # ~3e7 rows: blocks well above the pool's floor

    def rows(n):
        ctx.upload_packets(X0[:n])
        res = ctx.integrate_const_rows(step, n_iter, 25., narrow=True, resident=True)
        return res['store'], res['lengths']

    store, lengths = rows(len(X0))
    first = store.download(0, 5000)
    free_with_store = ctx.mem_info()[0]
    nbytes = store.nbytes
    store.free()
    assert ctx.mem_info()[0] >= free_with_store + nbytes*0.99          # pooled = free
    again, lengths2 = rows(len(X0) - 1000)           # a little smaller: takes the pooled blocks
    assert again.total == lengths2.sum() and np.array_equal(lengths2, lengths[:-1000])
    second = again.download(0, 5000)
    assert np.array_equal(first[0], second[0]) and np.array_equal(first[1], second[1])
    again.free()
    # with blocks waiting in the pool every other path works as before
    ctx.upload_packets(X0[:2000])
    dense = ctx.integrate_const(step, n_iter, 25., nrec=nsteps)['traj']
    c = coracle.integrate_const(f, X0[:2000], step, n_iter, 25., nrec=nsteps)
    assert np.array_equal(dense, c['traj'])


@pytest.mark.parametrize('sampler', ['numpy', 'device', 'pcg64'])
def test_input_run_splits_a_launch_group_whose_rows_do_not_fit(ctx, sampler, monkeypatch):
    """Input.run sizes its launch groups from an estimate of the rows per packet; when a group's
    rows do not fit in HBM after all (status NXC_ERR_NOMEM from the rows protocol) it is split
    in halves -- drawn again where the device draws -- and the catalogue comes out the same."""
    from nexoclom_amd import hip_api
    kw = dict(sampler='device', generator='pcg64') if sampler == 'pcg64' else dict(sampler=sampler)

    def run(limit):
        inputs = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
        inputs.options.endtime = type(inputs.options.endtime)(6000., 's')
        real = hip_api.Context.integrate_const_rows
        calls = []

        def tight(self, *a, **k):
            calls.append(self.n_packets)
            if limit and self.n_packets > limit:
                raise hip_api.HipError('nexoclom_hip error -6: no room for the rows of this '
                                       'launch group', hip_api.NXC_ERR_NOMEM)
            return real(self, *a, **k)
        monkeypatch.setattr(hip_api.Context, 'integrate_const_rows', tight)
        with contextlib.redirect_stdout(io.StringIO()):
            inputs.run(5000, packs_per_it=1000, seed=12, context=ctx, **kw)
        monkeypatch.setattr(hip_api.Context, 'integrate_const_rows', real)
        return inputs, calls
    whole, calls0 = run(0)
    split, calls1 = run(1500)
    if sampler == 'numpy':
        # host-drawn Outputs are launched as they become ready: the group sizes depend on timing
        assert sum(calls0) == 5000 and all(c % 1000 == 0 for c in calls0)
        done = [c for c in calls1 if c <= 1500]
        assert sum(done) == 5000 and set(done) == {1000}           # every launch that ran was split down
    else:
        assert calls0 == [5000] and calls1 == [5000, 2000, 1000, 1000, 3000, 1000, 2000, 1000, 1000]
    assert len(whole._catalogue) == len(split._catalogue) == 5
    for a, b in zip(whole._catalogue, split._catalogue):
        assert len(a.X) == len(b.X) and a.totalsource == b.totalsource
        for c in a.X.columns:
            assert np.array_equal(a.X[c].values, b.X[c].values), c
        for c in a.X0.columns:
            assert np.array_equal(a.X0[c].values, b.X0[c].values), c
