"""Host-side logic: inputfile parser, physical tables, X0 sampling (no GPU)."""
import os

import numpy as np
import pytest
from scipy import stats

import nexoclom_amd
from nexoclom_amd import Input, Output, PhotoRate, RadPresConst, SSObject, gValue, planet_dist
from nexoclom_amd.input_classes import InputError
from nexoclom_amd.Output import n_output_steps
from nexoclom_amd.source_distribution import xyz_from_lonlat
from tests import helpers as H

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_INPUTS = os.path.join(os.path.dirname(nexoclom_amd.__file__), 'inputfiles')


def test_photorate_known_answers():
    """tests/unit_tests/atomicdata/test_photolossrates.py:7-9 of the reference."""
    assert PhotoRate('Na', 1.5).rate.value == 3.2266666666666665e-06
    assert PhotoRate('Ca', 0.3).rate.value == 7.777777777777777e-4
    assert PhotoRate('X', 1.).rate.value == 1e-30


def test_mercury_setup_numbers():
    """Numbers obtained from the reference itself during the survey."""
    m = SSObject('mercury')
    assert m.object == 'Mercury' and m.type == 'Planet' and m.moons is None
    r, v = planet_dist(m, 1.3)
    assert float(r) == 0.35140097909804036
    assert abs(float(v)/9.730746760831499 - 1) < 1e-12
    R = m.radius.value*1e3
    assert abs(m.GM.value/R**3/(-1.51566332945409e-06) - 1) < 1e-14
    assert PhotoRate('Na', float(r)).rate.value == 5.8793685680196064e-05
    rp = RadPresConst('Na', float(r))
    assert len(rp.velocity) == 827 and np.all(np.diff(rp.velocity) > 0)
    assert abs(rp.accel.max()*1e5 - 361.09) < 0.01            # cm/s^2
    assert rp.velocity.min() == -50.6857 and rp.velocity.max() == 49.5196
    assert len(RadPresConst('Ca', 0.3).velocity) == 319
    assert len(RadPresConst('Mg', 0.3).velocity) == 179
    g = gValue('Na', 5891, 1.5)
    assert len(g.velocity) == 389 and np.all(np.diff(g.velocity) > 0)
    assert gValue('Zz', 1234).g.tolist() == [0., 0.]
    jup = SSObject('Jupiter')
    assert len(jup) == 5 and {x.object for x in jup.moons} == {'Io', 'Europa', 'Ganymede', 'Callisto'}


def test_xyz_from_lonlat_known_points():
    """tests/unit_tests/Initial_state/test_xyz_from_latlon.py of the reference: planet and
    satellite longitude conventions."""
    lon = np.array([0, np.pi/2, np.pi, 3*np.pi/2])
    p = xyz_from_lonlat(lon, np.zeros(4), True, 1.0)
    assert np.allclose(p.T, [[0, -1, 0], [1, 0, 0], [0, 1, 0], [-1, 0, 0]], atol=1e-15)
    s = xyz_from_lonlat(lon, np.zeros(4), False, 1.0)
    assert np.allclose(s.T, [[0, -1, 0], [-1, 0, 0], [0, 1, 0], [1, 0, 0]], atol=1e-15)
    lat = np.array([-np.pi/2, 0, np.pi/2])
    mer = xyz_from_lonlat(np.zeros(3), lat, True, 2.0)
    assert np.allclose(mer.T, [[0, 0, -2], [0, -2, 0], [0, 0, 2]], atol=1e-15)


def test_parser_defaults_and_rules():
    inp = Input(os.path.join(HERE, 'inputfiles', 'Gravity.input'))
    assert inp.geometry.planet.object == 'Mercury' and inp.geometry.taa == 3.14
    assert inp.geometry.type == 'geometry without starttime' and inp.geometry.phi is None
    assert inp.forces.gravity is True and inp.forces.radpres is False
    assert inp.surfaceinteraction.sticktype == 'constant'
    assert inp.surfaceinteraction.stickcoef == 1. and inp.surfaceinteraction.accomfactor is None
    assert inp.spatialdist.type == 'uniform' and inp.spatialdist.exobase == 1.
    assert [float(x) for x in inp.spatialdist.longitude] == [0., 2*np.pi]
    assert inp.speeddist.type == 'flat' and inp.speeddist.vprob.value == 4.
    assert inp.angulardist.type == 'isotropic'
    assert [float(x) for x in inp.angulardist.altitude] == [0., np.pi/2]
    assert inp.options.endtime.value == 20000. and inp.options.step_size == 30.
    assert inp.options.outeredge == 1e30 and inp.options.resolution is None
    assert inp.options.species == 'Na' and inp.options.lifetime.value == 0.

    reg = Input(os.path.join(HERE, 'inputfiles', 'Spatial.region.input'))
    assert reg.spatialdist.exobase == 2.1
    assert [float(x) for x in reg.spatialdist.latitude] == [0., 0.79]
    assert reg.speeddist.type == 'gaussian' and reg.speeddist.sigma.value == 0.5
    assert reg.angulardist.type == 'radial'
    assert reg.options.species == 'Na'            # capitalised (input_classes.py:1069)
    assert reg.options.outeredge == 12 and reg.options.step_size == 0.
    assert reg.options.resolution == 1e-4 and reg.options.lifetime.value == -7200.
    assert reg == Input(os.path.join(HERE, 'inputfiles', 'Spatial.region.input'))
    assert reg != inp
    with pytest.raises(FileNotFoundError):
        Input('/nonexistent.input')


def test_parser_errors(tmp_path):
    p = tmp_path / 'bad.input'
    p.write_text('SpatialDist.type = uniform\nSpeedDist.type = flat\noptions.endtime = 10\n'
                 'options.species = Na\n')
    with pytest.raises(InputError):
        Input(str(p))                              # no geometry.planet
    p.write_text('geometry.planet = Mercury\nSpatialDist.type = uniform\nSpeedDist.type = flat\n'
                 'SpeedDist.vprob = 1\noptions.endtime = 10\noptions.species = Na\n')
    with pytest.raises(InputError):
        Input(str(p))                              # flat without delv
    p.write_text('geometry.planet = Mercury\nSpatialDist.type = uniform\n'
                 'SpatialDist.latitude = 1, 0\nSpeedDist.type = flat\nSpeedDist.vprob = 1\n'
                 'SpeedDist.delv = 1\noptions.endtime = 10\noptions.species = Na\n')
    with pytest.raises(InputError):
        Input(str(p))                              # latitude[0] > latitude[1]


def test_n_output_steps_matches_reference_rule():
    assert n_output_steps(50000., 30.) == (1668, 1667)
    assert n_output_steps(10800., 30.) == (361, 360)
    assert n_output_steps(20000., 30.) == (668, 667)
    assert n_output_steps(100., 30.) == (5, 4)


def test_chunk_rule():
    inp = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
    assert inp.chunk_size() == 80467              # ceil(1024^3/1668/8), SURVEY.md section 3.1
    inp.options.step_size = 0
    assert inp.chunk_size() == 1000000


def test_x0_sampling_is_seed_deterministic_and_ordered():
    inp = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
    a = Output(inp, 5000, seed=1234, integrate=False, save=False)
    b = Output(inp, 5000, seed=1234, integrate=False, save=False)
    assert a.X0.equals(b.X0)
    cols = ['time', 'x', 'y', 'z', 'vx', 'vy', 'vz', 'frac']
    assert list(a.X0.columns) == cols + ['v', 'longitude', 'latitude', 'local_time', 'altitude',
                                         'azimuth']
    # independent restatement of the draw order (tests/helpers.sample_x0)
    assert np.array_equal(a.X0[cols].values, H.sample_x0(5000, 1234, 50000.))
    X = a.X0
    assert np.allclose(np.sqrt(X.x**2 + X.y**2 + X.z**2), 1.0)
    speed = np.sqrt(X.vx**2 + X.vy**2 + X.vz**2)*a.unit_km
    assert speed.min() >= 0.5 and speed.max() <= 4.5
    # outward launch: v . r > 0
    assert np.all(X.vx*X.x + X.vy*X.y + X.vz*X.z > 0)
    kw = a.forces_kwargs()
    assert kw['GM'] < 0 and kw['photo'] == 5.8793685680196064e-05 and len(kw['v_tab']) == 827


def test_uniform_surface_sampling_statistics():
    """KS tests in the spirit of the reference's test_spatial_distribution.py:95-143."""
    inp = Input(os.path.join(HERE, 'inputfiles', 'Spatial.region.input'))
    inp.options.step_size = 30.
    out = Output(inp, 100000, seed=7, integrate=False, save=False)
    lon, lat = out.X0.longitude.values, out.X0.latitude.values
    assert lon.min() >= 0 and lon.max() <= 3.14 and lat.min() >= 0 and lat.max() <= 0.79
    assert stats.kstest(lon, 'uniform', args=(0, 3.14)).pvalue > 1e-3
    assert stats.kstest(np.sin(lat), 'uniform', args=(0, np.sin(0.79))).pvalue > 1e-3
    r = np.sqrt(out.X0.x**2 + out.X0.y**2 + out.X0.z**2)
    assert np.allclose(r, 2.1)
    # radial launch: velocity parallel to position
    v = out.X0[['vx', 'vy', 'vz']].values
    p = out.X0[['x', 'y', 'z']].values
    assert np.allclose(np.cross(v, p), 0, atol=1e-12)
    assert out.loss_info.photo == 1/7200.


def test_surface_interaction_tables():
    """SurfaceInteraction.py:10-61 restated: temperature model, sticking law, v(T, p) table."""
    from nexoclom_amd.surface import SurfaceInteraction, bounce_config, surface_temperature
    inp = Input(os.path.join(HERE, 'inputfiles', 'Bounce.tempdep.input'))
    t = surface_temperature(inp.geometry, np.array([0., np.pi/2, np.pi, 0.]),
                            np.array([0., 0., 0., np.pi/3]))
    t1 = 600 + 125*(np.cos(1.3) - 1)/2
    assert np.allclose(t, [100 + t1, 100 + t1*np.cos(np.pi/2)**0.25, 100, 100 + t1*0.5**0.25])
    surf = SurfaceInteraction(inp)
    st = surf.stickcoef(np.array([0., np.pi]), np.array([0., 0.]))
    A = inp.surfaceinteraction.A
    assert np.allclose(st, np.clip(A[0]*np.exp(A[1]*np.array([100 + t1, 100.])) + A[2], 0, 1))
    assert surf.probgrid.shape == (201, 101) and np.all(np.diff(surf.probgrid, axis=1) >= 0)
    assert np.all(np.diff(surf.probgrid[:, 50]) > 0)          # hotter surface, faster atoms
    # median emission speed ~ 1.09 v_th for the v^3 exp(-v^2/vth^2) flux distribution
    vth = np.sqrt(2*surf.temperature*1.380649e-23/(22.98976928*1.66053906660e-27))/1e3
    assert np.allclose(surf.v_interp(surf.temperature, np.full(201, 0.5))/vth, 1.2958, atol=0.02)
    cfg = bounce_config(inp, -1.5e-6, 2440.53, 7)
    assert cfg['temp_dependent'] == 1 and cfg['accomfactor'] == 0.2 and len(cfg['tx']) == 205
    inp2 = Input(os.path.join(PKG_INPUTS, 'Na.mercury.bench.input'))
    assert bounce_config(inp2, -1.5e-6, 2440.53, 7) is None


def _write(tmp_path, text):
    p = tmp_path / 'case.input'
    p.write_text(text)
    return str(p)


BASE = ('SpatialDist.type = uniform\nSpeedDist.type = flat\nSpeedDist.vprob = 4.\n'
        'SpeedDist.delv = 4.\noptions.endtime = 10800.\noptions.species = Na\n')


def test_parser_geometry_variants(tmp_path):
    """The geometry cases of the reference's tests/test_data/inputfiles/Geometry.0[1-3].input and
    test_input_classes.py: planet with moons, start point, objects, phi, subsolarpoint, starttime."""
    inp = Input(_write(tmp_path, 'geometry.planet = Jupiter\ngeometry.StartPoint = Io\n'
                                 'geometry.objects = Jupiter, Io, Europa\ngeometry.phi = 1., 2.\n'
                                 'geometry.subsolarpoint = 3.14, 0\ngeometry.taa = 1.57\n' + BASE))
    g = inp.geometry
    assert g.planet.object == 'Jupiter' and g.startpoint == 'Io'
    assert {o.object for o in g.objects} == {'Jupiter', 'Io', 'Europa'}
    assert g.phi == (1., 2.) and g.subsolarpoint == (3.14, 0.) and g.taa == 1.57
    assert g.type == 'geometry without starttime'
    with pytest.raises(InputError):                      # wrong number of orbital positions
        Input(_write(tmp_path, 'geometry.planet = Jupiter\ngeometry.startpoint = Io\n'
                               'geometry.objects = Jupiter, Io, Europa\ngeometry.phi = 1.\n' + BASE))
    with pytest.raises(InputError):                      # moons but no phi
        Input(_write(tmp_path, 'geometry.planet = Jupiter\ngeometry.startpoint = Io\n' + BASE))
    with pytest.raises(ValueError):                      # start point not in the system
        Input(_write(tmp_path, 'geometry.planet = Mercury\ngeometry.startpoint = Io\n' + BASE))
    with pytest.raises(InputError):
        Input(_write(tmp_path, 'geometry.planet = Jupiter\ngeometry.startpoint = Io\n'
                               'geometry.objects = Jupiter, Moon\ngeometry.phi = 1.\n' + BASE))
    t = Input(_write(tmp_path, 'geometry.planet = Jupiter\ngeometry.StartPoint = Io\n'
                               'geometry.starttime = 2022-03-08T19:53:21\n' + BASE))
    assert t.geometry.type == 'geometry with starttime'
    assert t.geometry.time == '2022-03-08T19:53:21'


def test_parser_surface_and_distribution_variants(tmp_path):
    """SurfaceInteraction.0[1-6], Spatial, Speed and Angular variants of the reference's test
    inputfiles: defaults, clamping and the required-parameter errors."""
    geo = 'geometry.planet = Mercury\ngeometry.taa = 3.14\n'
    s = Input(_write(tmp_path, geo + 'surfaceinteraction.stickcoef = 1.7\n' + BASE))
    assert s.surfaceinteraction.stickcoef == 1 and s.surfaceinteraction.accomfactor is None
    s = Input(_write(tmp_path, geo + 'surfaceinteraction.stickcoef = -0.2\n'
                                     'surfaceinteraction.accomfactor = 0.4\n' + BASE))
    assert s.surfaceinteraction.stickcoef == 0 and s.surfaceinteraction.accomfactor == 0.4
    with pytest.raises(InputError):                      # partial sticking needs accomfactor
        Input(_write(tmp_path, geo + 'surfaceinteraction.stickcoef = 0.5\n' + BASE))
    s = Input(_write(tmp_path, geo + 'SurfaceInteraction.sticktype = temperature dependent\n'
                                     'SurfaceInteraction.accomfactor = 0.2\n' + BASE))
    assert s.surfaceinteraction.A == (1.57014, -0.006262, 0.1614157)
    s = Input(_write(tmp_path, geo + 'SurfaceInteraction.sticktype = temperature dependent\n'
                                     'SurfaceInteraction.accomfactor = 0.2\n'
                                     'SurfaceInteraction.A = 1, -0.01, 0.2\n' + BASE))
    assert s.surfaceinteraction.A == (1., -0.01, 0.2)
    with pytest.raises(InputError):
        Input(_write(tmp_path, geo + 'SurfaceInteraction.sticktype = temperature dependent\n'
                                     'SurfaceInteraction.accomfactor = 0.2\n'
                                     'SurfaceInteraction.A = 1, 2\n' + BASE))
    rest = 'options.endtime = 100\noptions.atom = ca\n'
    s = Input(_write(tmp_path, geo + 'SpatialDist.type = surface spot\nSpatialDist.longitude = 1.\n'
                                     'SpatialDist.latitude = 0.5\nSpatialDist.sigma = 0.3\n'
                                     'SpeedDist.type = maxwellian\nSpeedDist.temperature = 1200\n'
                                     'AngularDist.type = 2d\nAngularDist.altitude = 0.1, 9\n' + rest))
    assert s.spatialdist.sigma == 0.3 and s.speeddist.temperature.value == 1200.
    assert s.angulardist.type == '2d'
    assert [float(a) for a in s.angulardist.altitude] == [0.1, np.pi]     # clamped to pi
    assert s.options.species == 'Ca'
    s = Input(_write(tmp_path, geo + 'SpatialDist.type = uniform\nSpeedDist.type = sputtering\n'
                                     'SpeedDist.alpha = 3\nSpeedDist.beta = 0.7\nSpeedDist.U = 2.\n'
                                     'AngularDist.type = isotropic\nAngularDist.azimuth = 1, 0.5\n'
                                     + rest))
    assert s.speeddist.U.value == 2. and s.speeddist.alpha == 3.
    assert [float(a) for a in s.angulardist.azimuth] == [1., 0.5]
    with pytest.raises(InputError):
        Input(_write(tmp_path, geo + 'SpatialDist.type = surface spot\nSpatialDist.longitude = 1.\n'
                                     'SpeedDist.type = flat\nSpeedDist.vprob=1\nSpeedDist.delv=1\n'
                                     + rest))
    with pytest.raises(InputError):
        Input(_write(tmp_path, geo + 'SpatialDist.type = nonsense\nSpeedDist.type = flat\n'
                                     'SpeedDist.vprob=1\nSpeedDist.delv=1\n' + rest))


def test_host_samplers_for_other_sources():
    """maxwellian / sputtering speeds (global NumPy RNG, as the reference) and the surface spot:
    statistical sanity of the restated samplers."""
    import numpy.random as nprandom
    inp = Input(os.path.join(HERE, 'inputfiles', 'Gravity.input'))
    from nexoclom_amd.units import Quantity
    inp.speeddist.type = 'maxwellian'
    inp.speeddist.temperature = Quantity(1200., 'K')
    nprandom.seed(3)
    out = Output(inp, 50000, seed=1, integrate=False, save=False)
    v = out.X0.v.values*out.unit_km
    vth = np.sqrt(2*1200*1.380649e-23/(22.98976928*1.66053906660e-27))/1e3
    assert abs(np.median(v)/vth - 1.2958) < 0.02          # median of v^3 exp(-v^2/vth^2)
    inp.speeddist.type = 'sputtering'
    inp.speeddist.alpha, inp.speeddist.beta, inp.speeddist.U = 3., 0.7, Quantity(2., 'eV')
    out = Output(inp, 20000, seed=1, integrate=False, save=False)
    v = out.X0.v.values*out.unit_km
    assert 0.1 <= v.min() and v.max() <= 50 and 1.0 < np.median(v) < 6.0
    inp.spatialdist.type = 'surface spot'
    inp.spatialdist.longitude = Quantity(1.0, 'rad')
    inp.spatialdist.latitude = Quantity(0.3, 'rad')
    inp.spatialdist.sigma = Quantity(0.2, 'rad')
    inp.spatialdist.exobase = 1.0
    inp.speeddist.type = 'flat'
    inp.speeddist.vprob, inp.speeddist.delv = Quantity(2., 'km/s'), Quantity(1., 'km/s')
    out = Output(inp, 20000, seed=1, integrate=False, save=False)
    assert abs(np.median(out.X0.longitude) - 1.0) < 0.05


def test_tables_match_the_text_file_restatement():
    """Row a-5 beyond one (planet, taa, species) point: RadPresConst / gValue ARRAYS for Na, Ca,
    Mg at 0.3, 0.3514 and 1.5 au, PhotoRate for five species and planet_dist at five true
    anomalies equal tests/golden/g7_tables.npz, which oracle/make_table_golden.py computed from
    the reference's text data files with an independent restatement of its formulas."""
    g = np.load(os.path.join(HERE, 'golden', 'g7_tables.npz'))
    lines = {'Na': (3303, 5891, 5897), 'Ca': (2722, 4227, 4567), 'Mg': (2852,)}
    for sp, waves in lines.items():
        for k, a in enumerate(g['distances']):
            rp = RadPresConst(sp, a)
            assert np.array_equal(rp.velocity, g[f'{sp}_radpres_v'])
            np.testing.assert_allclose(rp.accel, g[f'{sp}_radpres_a{k}'], rtol=4e-16, atol=0)
            assert tuple(rp.wavelength) == waves
            for w in waves:
                gv = gValue(sp, w, a)
                assert np.array_equal(gv.velocity, g[f'{sp}_{w}_v'])
                assert np.array_equal(gv.g, g[f'{sp}_{w}_g{k}'])
    for sp in ('Na', 'Ca', 'Mg', 'K', 'O'):
        got = [PhotoRate(sp, a).rate.value for a in g['distances']]
        assert np.array_equal(got, g[f'{sp}_photo'])
    m = SSObject('Mercury')
    for taa, (r, vr) in zip(g['mercury_taa'], g['mercury_r_vr']):
        rr, vv = planet_dist(m, taa)
        assert float(rr) == r and float(vv) == vr
    assert m.GM.value/(m.radius.value*1e3)**3 == float(g['mercury_GM_R3'])


def test_los_geometry_shared_ladder_equals_per_spectrum_ladders():
    """compute_iteration.py:157-167 builds one geometric ladder per spectrum; los_geometry builds
    the longest once and finds each spectrum's length by search.  Same lengths, same rungs, for
    lines that leave the sphere, start outside it and never meet it (NaN root)."""
    from nexoclom_amd.LOSResult import SpacecraftData, ladder_to, los_geometry
    rng = np.random.default_rng(2)
    S = 200
    pos = rng.normal(size=(S, 3))*rng.uniform(0.5, 60, (S, 1))
    look = rng.normal(size=(S, 3))
    look /= np.linalg.norm(look, axis=1)[:, None]
    sc = SpacecraftData(*pos.T, *look.T)
    dphi = np.radians(1.0)
    for outeredge in (3., 25., 100.):
        lengths, longest = [], []
        for x_sc, bore in zip(pos, look):
            b = 2*np.sum(x_sc*bore)
            c = np.linalg.norm(x_sc)**2 - outeredge**2
            with np.errstate(invalid='ignore'):
                far = (-b + np.sqrt(b**2 - 4*1*c))/2
            rungs = ladder_to(far, np.sin(dphi), np.sin(dphi))
            lengths.append(len(rungs))
            longest = rungs if len(rungs) > len(longest) else longest
        _, got_lengths, got_ladder = los_geometry(sc.data, outeredge, dphi)
        assert np.array_equal(got_lengths, lengths)
        assert np.array_equal(got_ladder, np.array(longest))
        assert min(lengths) == 1 and max(lengths) > 100          # the NaN / inside cases occur


def test_output_frames_are_retyped_like_save_and_restore(tmp_path):
    """save()'s 32-bit down-cast and restore()'s up-cast (Output.py:528-543, 555-570) on whole
    frames, and the five image columns read back from an .npz without restoring the rest."""
    import pandas as pd
    from nexoclom_amd.Output import NARROW, WIDE, Output
    n = 1000
    rng = np.random.default_rng(4)
    frame = pd.DataFrame({'Index': np.arange(n, dtype=np.int64), 'x': rng.normal(size=n),
                          'frac': rng.random(n), 'flag': np.arange(n) % 2 == 0},
                         index=np.arange(n)*7)
    narrow = Output._recast(frame, NARROW)
    assert [str(t) for t in narrow.dtypes] == ['int32', 'float32', 'float32', 'bool']
    assert np.array_equal(narrow.index, frame.index)
    assert np.array_equal(narrow.x.values, frame.x.values.astype(np.float32))
    wide = Output.upcast(narrow)
    assert [str(t) for t in wide.dtypes] == ['int64', 'float64', 'float64', 'bool']
    assert np.array_equal(wide.x.values, frame.x.values.astype(np.float32).astype(np.float64))
    assert Output._recast(wide, NARROW) is not wide and Output._recast(wide, WIDE) is wide
    cols = {f'X.{c}': rng.normal(size=50).astype(np.float32) for c in ('x', 'y', 'z', 'vy', 'frac', 'vx')}
    path = str(tmp_path / 'out.npz')
    np.savez(path, aplanet=0.35, vrplanet_kms=9.7, **cols)
    samples, aplanet, vr = Output.image_columns(path)
    assert (aplanet, vr) == (0.35, 9.7) and len(samples) == 5
    for got, c in zip(samples, Output.IMAGE_COLS):
        assert got.dtype == np.float32 and np.array_equal(got, cols['X.' + c])


def test_windowed_host_draws_equal_slices_of_the_full_draw():
    """A rank that owns rows [a, b) of a chunk draws only those (WindowGenerator: PCG64.advance to
    stream position draw*n + a): every X0 column must be bit-identical to the same rows of the
    full chunk's X0, for the constant-step sources and for the variable-step run (whose launch
    times are one more draw in front)."""
    import contextlib
    import io
    from nexoclom_amd import Input, Output
    from nexoclom_amd.source_distribution import WindowGenerator
    pkg = os.path.join(os.path.dirname(nexoclom_amd.__file__), 'inputfiles')
    for name, variable in (('Na.mercury.bench.input', False), ('Na.mercury.bench.input', True),
                           ('Ca.isotropic.flat.input', False)):
        inputs = Input(os.path.join(pkg, name))
        if variable:
            inputs.options.step_size = 0.
            inputs.options.resolution = 1e-4
        assert WindowGenerator.windowable(inputs)
        n, seed = 2003, 77
        with contextlib.redirect_stdout(io.StringIO()):
            full = Output(inputs, n, seed=seed, integrate=False, save=False)
            for a, b in ((0, n), (0, 1), (5, 5), (1, 700), (1999, n), (640, 1311)):
                part = Output(inputs, n, seed=seed, window=(n, a, b), integrate=False, save=False)
                assert part.npackets == b - a and len(part.X0) == b - a
                assert list(part.X0.columns) == list(full.X0.columns)
                for c in full.X0.columns:
                    assert np.array_equal(part.X0[c].values, full.X0[c].values[a:b]), (name, c, a, b)
                assert np.array_equal(part.x0_soa(), full.x0_soa()[:, a:b])
    # the raw stream: five successive vectors, window by window
    n = 1000
    rng = np.random.default_rng(5)
    vectors = [rng.random(n) for _ in range(5)]
    win = WindowGenerator(5, n, 123, 877)
    for v in vectors:
        assert np.array_equal(win.random(754), v[123:877])
    # gaussian speeds cannot be windowed: such inputs draw whole chunks
    inputs = Input(os.path.join(pkg, 'Na.mercury.bench.input'))
    inputs.speeddist.type = 'gaussian'
    inputs.speeddist.sigma = type(inputs.speeddist.vprob)(0.5, 'km/s')
    assert not WindowGenerator.windowable(inputs)
    with pytest.raises(TypeError):
        win.standard_normal(10)
