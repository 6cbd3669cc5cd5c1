"""Initial packet states X0 on the host, draw-for-draw the reference's sequence.

What a seeded run must reproduce ("Seed parity") is the ORDER in which random
numbers are taken from ``output.randgen`` and the arithmetic that turns them into a state: release
time (variable-step runs only, drawn by Output before anything here) -> sin(latitude) -> longitude
-> speed -> sin(altitude) -> azimuth; each a whole ``npackets`` vector.  The reference spreads
this over initial_state/source_distribution.py:12-283, math/randomdeviates.py:8-83 and
math/distributions.py:7-21; here every kind of source is one small sampler function registered in
a table (``SURFACES``, ``SPEEDS``, ``DIRECTIONS``), and the three entry points the Output calls --
``surface_distribution``, ``speed_distribution``, ``angular_distribution`` -- look the sampler up,
run it and store the columns.  The order of generator calls is asserted by
tests/test_host.py::test_x0_sampling_is_seed_deterministic_and_in_reference_order.

Like the reference, the tabulated-density samplers ('maxwellian', 'sputtering', 'surface spot')
draw from the UNSEEDED process-global ``numpy.random`` (randomdeviates.py:33,63-65), so they are
statistically but not bitwise reproducible; the device sampler (nxc_packets_sample) covers them
with counter-based draws.  Sources that need the reference's pickled map files ('surface map',
'user defined') are out of scope.
"""
import numpy as np
import numpy.random as unseeded
from scipy.interpolate import interpn

from . import constants as const
from .input_classes import InputError

TWO_PI = 2*np.pi


# ---- a window of the seeded stream ---------------------------------------------------------------
class WindowGenerator:
    """Rows [a, b) of every ``n``-long uniform draw of ``numpy.random.default_rng(seed)``.

    The reference draws whole vectors, ``randgen.random(npackets)``, one after the other
    (Output.py:138-139; source_distribution.py:51-62,169-171,202-212), and each double consumes
    exactly one 64-bit output of the PCG64 stream: element i of draw j sits at stream position
    j*n + i.  A rank that owns only rows [a, b) of a chunk of n packets therefore jumps there with
    ``PCG64.advance`` (O(log) 128-bit multiplications) instead of drawing -- and discarding -- the
    whole chunk on every rank that overlaps it.  Bit-identical to slicing the full draw
    (tests/test_host.py).  Only ``random`` can be windowed: ``standard_normal`` (ziggurat) uses a
    data-dependent number of outputs, so gaussian sources draw whole chunks (``windowable``)."""

    def __init__(self, seed, n, a, b):
        if seed is None or not 0 <= a <= b <= n:
            raise ValueError('a window needs a seed and 0 <= a <= b <= n')
        self.n, self.a, self.b = int(n), int(a), int(b)
        self._start = np.random.PCG64(seed).state      # = default_rng(seed).bit_generator.state
        self._draws = 0

    def random(self, size):
        assert size == self.b - self.a, 'a windowed generator draws windows of whole vectors'
        engine = np.random.PCG64(0)
        engine.state = self._start
        engine.advance(self._draws*self.n + self.a)
        self._draws += 1
        return np.random.Generator(engine).random(size)

    def standard_normal(self, size):
        raise TypeError('standard_normal cannot be windowed (see WindowGenerator.windowable)')

    @staticmethod
    def windowable(inputs):
        """Whether every seeded draw these inputs make is a ``random(npackets)`` vector."""
        gaussian = inputs.speeddist.type.lower() == 'gaussian' and inputs.speeddist.sigma != 0.
        return not gaussian


class LaunchTable(dict):
    """X0 while it is being drawn: plain arrays by column name.  (Every column inserted into a
    pandas frame costs about a millisecond with the GIL held -- a third of the time it takes to
    draw an Output of 8e4 packets, and what kept several Outputs from being drawn side by side;
    Output builds the frame once, from the finished table.)  Scalars are broadcast like pandas
    broadcasts them."""

    def __init__(self, n):
        super().__init__()
        self.n = int(n)

    def __setitem__(self, name, value):
        column = np.asarray(value)
        if column.ndim == 0:
            column = np.full(self.n, column)
        super().__setitem__(name, column)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def frame(self, columns):
        import pandas as pd
        return pd.DataFrame({c: self[c] for c in columns}, copy=False)


# ---- geometry helpers ---------------------------------------------------------------------------
def xyz_from_lonlat(lon, lat, isplan, exobase):
    """Surface point of longitude/latitude on the sphere r = exobase, as a (3, n) array
    (source_distribution.py:12-34).  Planet: longitude 0 is the subsolar point (0,-1,0), pi/2 the
    dusk terminator (+x).  Satellite: longitude 0 is the sub-planet point, pi/2 the leading
    point (-x)."""
    handed = 1.0 if isplan else -1.0
    xyz = np.array([handed * exobase * np.sin(lon) * np.cos(lat),
                    -exobase * np.cos(lon) * np.cos(lat),
                    exobase * np.sin(lat)])
    assert np.all(np.isfinite(xyz)), 'Non-Finite values of X0'
    return xyz


def _unit_rows(vectors):
    return vectors/np.linalg.norm(vectors, axis=1)[:, np.newaxis]


def local_frame(x, y, z):
    """Unit vectors (radial, east, north), each (n, 3), of the launch points."""
    nothing = np.zeros_like(z)
    radial = np.array([x, y, z]).transpose()
    east = np.array([y, -x, nothing]).transpose()
    north = np.array([-z*x, -z*y, x**2+y**2]).transpose()
    return _unit_rows(radial), _unit_rows(east), _unit_rows(north)


def _unit_columns(a, b, c):
    """(a, b, c)/|(a, b, c)| component by component.  The norm is formed like np.linalg.norm
    forms it for the rows of an (n, 3) array -- sqrt((a*a + b*b) + c*c) -- so the columns carry
    the bits `_unit_rows` would give, without the strided (n, 3) temporaries."""
    length = np.sqrt((a*a + b*b) + c*c)
    return a/length, b/length, c/length


def local_frame_columns(x, y, z):
    """`local_frame` as three tuples of 1-D arrays (bit-identical components)."""
    return (_unit_columns(x, y, z), _unit_columns(y, -x, np.zeros_like(z)),
            _unit_columns(-z*x, -z*y, x**2+y**2))


def _ascending(first, second):
    """An angular range that runs through 2 pi is unwrapped so that it ascends."""
    return (first, second) if first <= second else (first, second + TWO_PI)


# ---- densities and deviates (global generator) ------------------------------------------------
def _mass_kg(species):
    return const.ATOMIC_MASS[species] * const.AMU


def sputdist(velocity, U_eV, alpha, beta, species):
    """Sputtering flux density over velocity [km/s], peak-normalised (distributions.py:7-13):
    v^(2 beta + 1) / (v^2 + v_b^2)^alpha with v_b the speed of binding energy U."""
    v_b = np.sqrt(2*U_eV*const.EV/_mass_kg(species)) / 1e3
    density = velocity**(2*beta+1) / (velocity**2 + v_b**2)**alpha
    return density / np.max(density)


def MaxwellianDist(velocity, temperature, species):
    """Maxwellian FLUX density over velocity [km/s], peak-normalised (distributions.py:16-21)."""
    vth2 = 2*temperature*const.K_B/_mass_kg(species) / 1e6
    density = velocity**3 * np.exp(-velocity**2/vth2)
    return density / np.max(density)


def density_cdf(x, f_x):
    """(cdf, grid) of a density tabulated on x: running sum shifted to start at 0 and scaled to
    end at 1, on an even grid over x's range (randomdeviates.py:29-32)."""
    grid = np.linspace(x.min(), x.max(), f_x.shape[0])
    cdf = f_x.cumsum()
    cdf -= cdf.min()
    cdf /= cdf.max()
    return cdf, grid


def random_deviates_1d(x, f_x, num):
    """Inverse-CDF deviates of the density f_x tabulated on x (randomdeviates.py:8-33)."""
    cdf, grid = density_cdf(x, f_x)
    return np.interp(unseeded.rand(num), cdf, grid)


def random_deviates_2d(fdist, x0, y0, num):
    """Accept/reject deviates of a density map on (x0, y0): rounds of ``num`` candidate points
    under a box of height max(fdist) until ``num`` are accepted (randomdeviates.py:36-83)."""
    span_x, span_y = x0.max() - x0.min(), y0.max() - y0.min()
    ceiling = fdist.max()
    axes = (np.linspace(x0.min(), x0.max(), fdist.shape[0]),
            np.linspace(y0.min(), y0.max(), fdist.shape[1]))
    kept_x, kept_y = [], []
    while len(kept_x) < num:
        cand_x = unseeded.rand(num)*span_x + x0.min()
        cand_y = unseeded.rand(num)*span_y + y0.min()
        height = unseeded.rand(num)*ceiling
        under = height < interpn(axes, fdist, (cand_x, cand_y))
        kept_x.extend(cand_x[under])
        kept_y.extend(cand_y[under])
    return np.array(kept_x[:num]), np.array(kept_y[:num])


# ---- where packets start: (lon, lat) ------------------------------------------------------------
def spot_density_map(lon0, lat0, sigma):
    """exp(-angular distance / sigma) from the spot centre on a 1-degree (lon, lat) grid
    (source_distribution.py:96-113; the reference's map uses -sin(lat) for z, kept)."""
    centre = (np.sin(lon0)*np.cos(lat0), -np.cos(lon0)*np.cos(lat0), np.sin(lat0))
    lon = np.linspace(0, TWO_PI, 361)
    lat = np.linspace(-np.pi/2, np.pi/2, 181)
    cosine = (np.outer(np.sin(lon), np.cos(lat))*centre[0]
              + -np.outer(np.cos(lon), np.cos(lat))*centre[1]
              + -np.outer(np.ones_like(lon), np.sin(lat))*centre[2])
    return lon, lat, np.exp(-np.arccos(np.clip(cosine, -1, 1))/sigma)


def _surface_uniform(out, sd):
    n, rng = out.npackets, out.randgen
    s0, s1 = np.sin(sd.latitude[0]), np.sin(sd.latitude[1])
    lat = np.arcsin(s0 + (s1-s0) * rng.random(n))
    w0, w1 = _ascending(float(sd.longitude[0]), float(sd.longitude[1]))
    lon = (w0 + (w1-w0) * rng.random(n)) % TWO_PI
    return lon, lat


def _surface_spot(out, sd):
    lon, lat, density = spot_density_map(float(sd.longitude), float(sd.latitude),
                                         float(sd.sigma))
    return random_deviates_2d(density, lon, lat, out.npackets)


def _surface_map(out, sd):
    raise NotImplementedError("surface-map sources need the reference's pickled map files; "
                              'out of scope (SURVEY.md section 2)')


SURFACES = {'uniform': _surface_uniform, 'surface spot': _surface_spot,
            'surface map': _surface_map}


def surface_distribution(outputs):
    """Columns x, y, z, longitude, latitude, local_time of X0 (source_distribution.py:37-134)."""
    sd = outputs.inputs.spatialdist
    assert sd.type in SURFACES, "Can't get here"
    lon, lat = SURFACES[sd.type](outputs, sd)
    # The reference always uses the planet convention (:126).  With a moon as start point (our
    # extension) the satellite convention of xyz_from_lonlat applies.
    geo = outputs.inputs.geometry
    x, y, z = xyz_from_lonlat(lon, lat, geo.planet.object == geo.startpoint, sd.exobase)
    X0 = outputs.X0
    X0['x'], X0['y'], X0['z'] = x, y, z
    X0['longitude'], X0['latitude'] = lon, lat
    X0['local_time'] = (lon * 12/np.pi + 12) % 24


# ---- how fast: speed in km/s --------------------------------------------------------------------
def _speed_gaussian(out, vd, species):
    if vd.sigma == 0.:
        return np.zeros(out.npackets) + vd.vprob.value
    return out.randgen.standard_normal(out.npackets) * vd.sigma.value + vd.vprob.value


def _speed_flat(out, vd, species):
    return out.randgen.random(out.npackets)*2*vd.delv.value + vd.vprob.value - vd.delv.value


def tabulated_speed_density(vd, species):
    """(velocity grid [km/s], flux density) of the 'sputtering' and 'maxwellian' speed
    distributions on the reference's 5000-point grids (source_distribution.py:148-168)."""
    if vd.type == 'sputtering':
        grid = np.linspace(.1, 50, 5000)
        return grid, sputdist(grid, vd.U.value, vd.alpha, vd.beta, species)
    assert vd.type == 'maxwellian' and vd.temperature != 0, 'Not implemented yet'
    v_th = np.sqrt(2*vd.temperature.value*const.K_B/_mass_kg(species)) / 1e3
    grid = np.linspace(0.1, v_th*5, 5000)
    return grid, MaxwellianDist(grid, vd.temperature.value, species)


def _speed_tabulated(out, vd, species):
    return random_deviates_1d(*tabulated_speed_density(vd, species), out.npackets)


def _speed_from_file(out, vd, species):
    raise InputError('speed_distribution', 'user-defined speed files are out of scope')


SPEEDS = {'gaussian': _speed_gaussian, 'flat': _speed_flat, 'sputtering': _speed_tabulated,
          'maxwellian': _speed_tabulated, 'user defined': _speed_from_file}


def speed_distribution(outputs):
    """Column v of X0, in R/s (source_distribution.py:137-189)."""
    vd = outputs.inputs.speeddist
    kind = vd.type.lower() if vd.type.lower() == 'gaussian' else vd.type
    assert kind in SPEEDS, 'Distribtuion does not exist'
    v0 = SPEEDS[kind](outputs, vd, outputs.inputs.options.species) / outputs.unit_km
    outputs.X0['v'] = v0
    assert np.all(np.isfinite(v0)), 'Infinite values for v0'
    return v0


# ---- which way: altitude above the local horizon, azimuth from north through east ---------------
def _direction_radial(out, ad):
    n = out.npackets
    return np.zeros(n) + np.pi/2., np.zeros(n)


def _direction_isotropic(out, ad):
    n, rng = out.npackets, out.randgen
    s0, s1 = np.sin(ad.altitude[0]), np.sin(ad.altitude[1])
    alt = np.arcsin(rng.random(n) * (s1 - s0) + s0)
    a0, a1 = float(ad.azimuth[0]), float(ad.azimuth[1])
    lo, hi = (a0, a1) if a0 <= a1 else (a1, a0 + TWO_PI)     # the reference's unwrap (:209)
    return alt, lo + (hi-lo)*rng.random(n)


def _direction_planar(out, ad):
    c0, c1 = np.cos(ad.altitude[0]), np.cos(ad.altitude[1])
    return np.arccos(out.randgen.random(out.npackets) * (c1 - c0) + c0), None


DIRECTIONS = {'radial': _direction_radial, 'isotropic': _direction_isotropic,
              '2d': _direction_planar}


def angular_distribution(outputs):
    """Columns vx, vy, vz, altitude, azimuth of X0 (source_distribution.py:192-283): the speed
    along (cos alt cos az) north + (cos alt sin az) east + (sin alt) radial; for '2d' sources the
    motion stays in the equatorial plane."""
    ad = outputs.inputs.angulardist
    if ad.type == 'none':
        return
    assert ad.type in DIRECTIONS, 'Angular Distribution not defined.'
    alt, az = DIRECTIONS[ad.type](outputs, ad)
    X0 = outputs.X0
    x, y, z = (np.asarray(X0[c]) for c in ('x', 'y', 'z'))
    speed = np.asarray(X0['v'])
    up, level = np.sin(alt), np.cos(alt)
    if ad.type == '2d':
        outward = _unit_rows(np.array([x, y]).transpose())
        along = _unit_rows(np.array([y, -x]).transpose())
        heading = level[:, np.newaxis]*along + up[:, np.newaxis]*outward
        assert np.all(np.isclose(np.sum(heading**2, axis=1), 1))
        X0['vx'], X0['vy'] = heading[:, 0] * speed, heading[:, 1] * speed
        X0['vz'] = np.zeros((outputs.npackets, ))
        X0['altitude'], X0['azimuth'] = alt, 0
        X0['v_radial'] = up * X0['v']
        X0['v_east'] = np.sqrt(X0['v']**2 - X0['v_radial']**2)
        X0['v_north'] = 0
        return
    radial, east, north = local_frame_columns(x, y, z)
    to_north, to_east = level * np.cos(az), level * np.sin(az)
    for name, r, e, n in zip(('vx', 'vy', 'vz'), radial, east, north):
        X0[name] = ((to_north*n + to_east*e) + up*r) * speed
    X0['altitude'], X0['azimuth'] = alt, az
