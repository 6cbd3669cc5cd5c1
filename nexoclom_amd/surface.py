"""Surface interaction set-up for re-emitted packets (host side).

Re-statement of initial_state/surface_temperature.py:4-19 and
particle_tracking/SurfaceInteraction.py:10-61 of the reference: the Mercury surface temperature
model, the temperature-dependent sticking coefficient, and the table of thermally accommodated
emission speeds v(T, probability) with its interpolating bicubic spline (scipy
RectBivariateSpline = FITPACK).  ``bounce_tables()`` exports the spline's knots and coefficients so
that the HIP kernel evaluates the same spline (de Boor) at every impact.
"""
import numpy as np
from scipy import interpolate

from . import constants as const
from .source_distribution import MaxwellianDist


def surface_temperature(geometry, longitude, latitude, t0=100., t1=None, n=.25):
    """surface_temperature.py:4-19 (Mercury only): t0 on the night side,
    t0 + t1 |cos(lon) cos(lat)|^n on the day side, t1 = 600 + 125 (cos(taa) - 1)/2."""
    if geometry.startpoint != 'Mercury':
        raise NotImplementedError('surface temperature is only defined for Mercury')
    if t1 is None:
        t1 = 600. + 125*(np.cos(float(geometry.taa)) - 1)/2.
    longitude = np.asarray(longitude, dtype=float)
    latitude = np.asarray(latitude, dtype=float)
    t_surf = np.zeros_like(longitude) + t0
    mask = (longitude <= np.pi/2) | (longitude >= 3*np.pi/2)
    t_surf[mask] = t0 + t1*np.abs(np.cos(longitude[mask]) * np.cos(latitude[mask]))**n
    return t_surf


def day_side_t1(geometry):
    return 600. + 125*(np.cos(float(geometry.taa)) - 1)/2.


class SurfaceInteraction:
    """SurfaceInteraction.py:10-61: ``stickcoef(lon, lat)`` for temperature-dependent sticking
    and ``v_interp(T, p)`` [km/s] for accommodation (when accomfactor != 0)."""

    def __init__(self, inputs, nt=201, nv=101, nprob=101):
        sint = inputs.surfaceinteraction
        self.inputs = inputs
        if sint.sticktype == 'temperature dependent':
            A = sint.A

            def stickcoef(lon, lat):
                tsurf = surface_temperature(inputs.geometry, lon, lat)
                coef = A[0] * np.exp(A[1]*tsurf) + A[2]
                coef[coef > 1.] = 1.
                coef[coef < 0.] = 0.
                return coef
            self.stickcoef = stickcoef
        elif sint.sticktype == 'surface map':
            assert 0
        self.spline = None
        if sint.accomfactor != 0:
            longitude = np.arange(361)*np.pi/180.
            latitude = np.arange(181)*np.pi/180. - np.pi/2.
            longrid, latgrid = np.meshgrid(longitude, latitude)
            tsurf = surface_temperature(inputs.geometry, longrid.flatten(), latgrid.flatten())
            temperature = np.linspace(min(tsurf), max(tsurf), nt)
            mass = const.ATOMIC_MASS[inputs.options.species]*const.AMU
            v_temp = np.sqrt(2*temperature*const.K_B/mass)/1e3            # km/s
            probability = np.linspace(0, 1, nprob)
            probgrid = np.ndarray((nt, nprob))
            for i, t in enumerate(temperature):
                vrange = np.linspace(0, v_temp[i]*3, nv)
                f_v = MaxwellianDist(vrange, t, inputs.options.species)
                cumdist = f_v.cumsum()
                cumdist -= cumdist.min()
                cumdist /= cumdist.max()
                probgrid[i, :] = np.interp(probability, cumdist, vrange)
            self.spline = interpolate.RectBivariateSpline(temperature, probability, probgrid)
            self.v_interp = self.spline.ev
            self.probgrid = probgrid
            self.temperature = temperature
            self.probability = probability

    def bounce_tables(self):
        """(tx, ty, coef[nx-4, ny-4]) of the bicubic spline, or three one-element dummies when
        there is no accommodation."""
        if self.spline is None:
            return np.zeros(8), np.zeros(8), np.zeros((4, 4))
        tx, ty, c = self.spline.tck
        return (np.ascontiguousarray(tx), np.ascontiguousarray(ty),
                np.ascontiguousarray(c.reshape(len(tx)-4, len(ty)-4)))


def bounce_config(inputs, GM, unit_km, seed):
    """Everything the kernels need to re-emit a packet that hit the surface
    (particle_tracking/bouncepackets.py:39-100), as keyword arguments of
    hip_api.Context.set_bounce; None when packets simply stick (stickcoef == 1)."""
    sint = inputs.surfaceinteraction
    if sint.sticktype == 'constant' and sint.stickcoef == 1.:
        return None
    surf = SurfaceInteraction(inputs)
    tx, ty, coef = surf.bounce_tables()
    tdep = sint.sticktype == 'temperature dependent'
    return dict(GM=float(GM), unit_km=float(unit_km),
                accomfactor=float(sint.accomfactor or 0.0),
                temp_dependent=int(tdep),
                stickcoef=0.0 if tdep else float(sint.stickcoef),
                A=tuple(sint.A) if tdep else (0., 0., 0.),
                t0=100., t1=float(day_side_t1(inputs.geometry)), tpow=0.25,
                tx=tx, ty=ty, coef=coef, seed=0 if seed is None else int(seed),
                surf=surf)
