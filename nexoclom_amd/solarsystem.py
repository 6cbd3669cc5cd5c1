"""Solar-system bodies and the planet's heliocentric distance / radial velocity (host side).

Drop-in for solarsystem/SSObject.py:28-100 and solarsystem/planet_dist.py:9-74 of the reference
without astropy or pandas pickles.  The table of bodies is read once from
data/planetary_constants.csv (built by tools/make_data.py from the reference's text file
``PlanetaryConstants.dat``) into a plain dict keyed by case-folded name; ``SSObject`` is a view of
one record, moons are found through the ``orbits`` column.

planet_dist: the distance is the conic r = a (1 - e^2) / (1 + e cos nu).  The reference does not
differentiate that analytically for the radial velocity: it samples one orbit at 1001 mean
anomalies, maps them to true anomaly with the third-order equation-of-centre series, takes first
differences of r over the equal time steps and interpolates the result at the requested true
anomaly (planet_dist.py:36-67).  Drop-in parity needs THAT number (the Doppler shift of every
g-value lookup depends on it; the analytic sqrt(GM/p) e sin nu gives 9.692 km/s for
Mercury at nu = 1.3 where the recipe gives 9.731), so the same recipe is evaluated here -- once per orbit, cached as a
(true anomaly, dr/dt) curve -- and checked against the values the survey recorded from the
reference (Mercury at nu = 1.3 -> 0.35140097909804036 au,
9.730746760831499 km/s).
"""
import csv
import functools
import os

import numpy as np

from . import constants as const
from .units import Quantity, register_unit

_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data',
                      'planetary_constants.csv')
_NUMERIC = ('radius', 'mass', 'a', 'e', 'tilt', 'rot_period', 'orb_period')


@functools.lru_cache(maxsize=None)
def _bodies():
    """{case-folded name: record}; records keep the file order in ``rank``."""
    table = {}
    with open(_TABLE, newline='') as handle:
        for rank, row in enumerate(csv.DictReader(handle)):
            record = {key: float(row[key]) for key in _NUMERIC}
            record.update(name=row['Object'], orbits=row['orbits'], rank=rank)
            table[row['Object'].casefold()] = record
    return table


def _satellites(name):
    return [rec['name'] for rec in sorted(_bodies().values(), key=lambda r: r['rank'])
            if rec['orbits'] == name]


class SSObject:
    """A Solar System object (SSObject.py:30-73).  ``GM`` is NEGATIVE, -G M in m^3/s^2 (:53): the
    force model multiplies it straight into r/|r|^3."""

    _KIND = {'Milky Way': ('Star', 'km'), 'Sun': ('Planet', 'au')}

    def __init__(self, obj):
        record = _bodies().get(str(obj).casefold())
        self.object = record['name'] if record else None
        if record is None:
            print(f'Object {obj} does not exist in table.')
            return
        self.orbits = record['orbits']
        self.radius = Quantity(record['radius'], 'km')
        self.mass = Quantity(record['mass'], 'kg')
        self.e = record['e']
        self.tilt = Quantity(record['tilt'], 'deg')
        self.rotperiod = Quantity(record['rot_period'], 'h')
        self.orbperiod = Quantity(record['orb_period'], 'd')
        self.GM = Quantity(-record['mass']*const.G, 'm3/s2')
        self.moons = [SSObject(name) for name in _satellites(self.object)] or None
        self.type, length_unit = self._KIND.get(self.orbits, ('Moon', 'km'))
        self.a = Quantity(record['a'], length_unit)
        register_unit('R_' + self.object, 'length', record['radius']*1e3)

    def __len__(self):
        return 1 + len(self.moons or ())

    def __eq__(self, other):
        return self.object == getattr(other, 'object', other)

    def __hash__(self):
        return hash((self.object,))

    def __repr__(self):
        return f'SSObject({self.object})'


# ---- heliocentric distance and radial velocity ------------------------------------------------------
def conic_radius(a, e, true_anomaly):
    return a*(1 - e**2)/(1 + e*np.cos(true_anomaly))


def equation_of_centre(mean_anomaly, e):
    """True anomaly from mean anomaly, series to third order in e (planet_dist.py:48-51)."""
    m = mean_anomaly
    return (m + (2*e - e**3/4)*np.sin(m) + 5/4 * e**2 * np.sin(2*m)
            + 13/12 * e**3 * np.sin(3*m))


@functools.lru_cache(maxsize=None)
def _radial_velocity_curve(a, e, period_s, samples=1000):
    """(true anomaly, dr/dt [au/s]) along one orbit: ``samples`` equal time steps from
    perihelion plus one step before it, first differences assigned to the later point."""
    t = np.linspace(0, 1, samples)*period_s
    t = np.concatenate([np.array([t[0] - t[1]]), t])
    m = np.linspace(0, 2*np.pi, samples)
    m = np.concatenate([np.array([m[0] - m[1]]), m])
    nu = equation_of_centre(m, e)
    r = conic_radius(a, e, nu)
    return nu[1:], (r[1:] - r[:-1])/(t[1:] - t[:-1])


def planet_dist(planet_, taa=None, time=None):
    """(distance [au], radial velocity relative to the Sun [km/s]) of a planet at true anomaly
    ``taa`` [rad] (planet_dist.py:9-74).  ``time`` (the reference's SPICE route) is out of
    scope."""
    if not isinstance(planet_, (SSObject, str)):
        raise TypeError('solarsystemMB.planet_dist', 'Must give a SSObject or a object name.')
    planet = SSObject(planet_) if isinstance(planet_, str) else planet_
    if planet.object is None:
        return None
    if time is not None:
        raise NotImplementedError('planet_dist(time=...) needs SPICE kernels: out of scope')
    if taa is None:
        print('Neither a time nor a true anomaly was given.')
        return None
    a, e, nu = planet.a.value, planet.e, float(taa)
    if not e > 0:
        return Quantity(a, 'au'), Quantity(0., 'km/s')
    curve_nu, drdt = _radial_velocity_curve(a, e, planet.orbperiod.value*86400.)
    v_r = np.interp(nu, curve_nu, drdt*(const.AU_M/1e3))
    return Quantity(conic_radius(a, e, nu), 'au'), Quantity(v_r, 'km/s')
