"""Surface interaction set-up for re-emitted packets (host side).

What the kernels need when a packet that hits the surface is not simply absorbed
(particle_tracking/bouncepackets.py:39-100): the Mercury surface-temperature model
(initial_state/surface_temperature.py:4-19), the temperature-dependent sticking coefficient and
the table of thermally accommodated emission speeds v(T, probability) with its interpolating
bicubic spline (particle_tracking/SurfaceInteraction.py:10-61: scipy RectBivariateSpline =
FITPACK).  ``bounce_tables()`` exports the spline's knots and coefficients so that the HIP kernel
evaluates the very same spline (de Boor) at every impact.
"""
import numpy as np
from scipy import interpolate

from . import constants as const
from .source_distribution import MaxwellianDist, density_cdf

NIGHT_SIDE_K = 100.


def day_side_t1(geometry):
    """Sub-solar excess temperature [K] at the planet's true anomaly."""
    return 600. + 125*(np.cos(float(geometry.taa)) - 1)/2.


def surface_temperature(geometry, longitude, latitude, t0=NIGHT_SIDE_K, t1=None, n=.25):
    """Mercury only (surface_temperature.py:4-19): t0 on the night side; on the day side (within
    90 degrees of the sub-solar longitude) t0 + t1 |cos(lon) cos(lat)|^n."""
    if geometry.startpoint != 'Mercury':
        raise NotImplementedError('surface temperature is only defined for Mercury')
    excess = day_side_t1(geometry) if t1 is None else t1
    lon = np.asarray(longitude, dtype=float)
    lat = np.asarray(latitude, dtype=float)
    day = (lon <= np.pi/2) | (lon >= 3*np.pi/2)
    temperature = np.full_like(lon, t0)
    temperature[day] = t0 + excess*np.abs(np.cos(lon[day]) * np.cos(lat[day]))**n
    return temperature


class SurfaceInteraction:
    """SurfaceInteraction.py:10-61: ``stickcoef(lon, lat)`` for temperature-dependent sticking
    and ``v_interp(T, p)`` [km/s] for accommodation (when accomfactor != 0): the speed below which
    a fraction p of a Maxwellian flux at temperature T is emitted, tabulated on ``nt``
    temperatures spanning the planet's surface and ``nprob`` probabilities."""

    def __init__(self, inputs, nt=201, nv=101, nprob=101):
        spec = inputs.surfaceinteraction
        self.inputs = inputs
        assert spec.sticktype != 'surface map', 'sticking maps are out of scope'
        if spec.sticktype == 'temperature dependent':
            self.stickcoef = self._sticking_law(inputs.geometry, spec.A)
        self.spline = None
        if spec.accomfactor != 0:
            self._tabulate(inputs, nt, nv, nprob)

    @staticmethod
    def _sticking_law(geometry, A):
        def stickcoef(lon, lat):
            warm = surface_temperature(geometry, lon, lat)
            return np.clip(A[0] * np.exp(A[1]*warm) + A[2], 0., 1.)
        return stickcoef

    def _tabulate(self, inputs, nt, nv, nprob):
        species = inputs.options.species
        lon, lat = np.meshgrid(np.arange(361)*np.pi/180., np.arange(181)*np.pi/180. - np.pi/2.)
        everywhere = surface_temperature(inputs.geometry, lon.flatten(), lat.flatten())
        self.temperature = np.linspace(min(everywhere), max(everywhere), nt)
        self.probability = np.linspace(0, 1, nprob)
        mass = const.ATOMIC_MASS[species]*const.AMU
        thermal = np.sqrt(2*self.temperature*const.K_B/mass)/1e3            # km/s
        self.probgrid = np.ndarray((nt, nprob))
        for row, (kelvin, v_th) in enumerate(zip(self.temperature, thermal)):
            speeds = np.linspace(0, v_th*3, nv)
            cdf, grid = density_cdf(speeds, MaxwellianDist(speeds, kelvin, species))
            self.probgrid[row, :] = np.interp(self.probability, cdf, grid)
        self.spline = interpolate.RectBivariateSpline(self.temperature, self.probability,
                                                      self.probgrid)
        self.v_interp = self.spline.ev

    def bounce_tables(self):
        """(tx, ty, coef[nx-4, ny-4]) of the bicubic spline, or zero-filled dummies when there is
        no accommodation."""
        if self.spline is None:
            return np.zeros(8), np.zeros(8), np.zeros((4, 4))
        tx, ty, c = self.spline.tck
        return (np.ascontiguousarray(tx), np.ascontiguousarray(ty),
                np.ascontiguousarray(c.reshape(len(tx)-4, len(ty)-4)))


def bounce_config(inputs, GM, unit_km, seed):
    """Keyword arguments of hip_api.Context.set_bounce for these inputs; None when packets simply
    stick (stickcoef == 1)."""
    spec = inputs.surfaceinteraction
    if spec.sticktype == 'constant' and spec.stickcoef == 1.:
        return None
    surf = SurfaceInteraction(inputs)
    tx, ty, coef = surf.bounce_tables()
    by_temperature = spec.sticktype == 'temperature dependent'
    return dict(GM=float(GM), unit_km=float(unit_km),
                accomfactor=float(spec.accomfactor or 0.0),
                temp_dependent=int(by_temperature),
                stickcoef=0.0 if by_temperature else float(spec.stickcoef),
                A=tuple(spec.A) if by_temperature else (0., 0., 0.),
                t0=NIGHT_SIDE_K, t1=float(day_side_t1(inputs.geometry)), tpow=0.25,
                tx=tx, ty=ty, coef=coef, seed=0 if seed is None else int(seed),
                surf=surf)
