"""Solar-system constants and heliocentric distance / radial velocity (host side).

Re-statement without astropy of solarsystem/SSObject.py:28-100 and
solarsystem/planet_dist.py:9-74 of the reference.
"""
import functools
import os

import numpy as np
import pandas as pd

from . import constants as const
from .units import Quantity, register_unit

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


@functools.lru_cache(maxsize=None)
def _constants():
    return pd.read_csv(os.path.join(_DATA, 'planetary_constants.csv'))


class SSObject:
    """Solar System object (SSObject.py:30-73).  ``GM`` is NEGATIVE (-M*G, :53) in m^3/s^2."""

    def __init__(self, obj):
        constants = _constants()
        row = constants.loc[constants.Object.apply(lambda x: x.casefold()) == obj.casefold()]
        if len(row) == 1:
            row = row.iloc[0]
            self.object = row.Object
            self.orbits = row.orbits
            self.radius = Quantity(row.radius, 'km')
            self.mass = Quantity(row.mass, 'kg')
            self.e = float(row.e)
            self.tilt = Quantity(row.tilt, 'deg')
            self.rotperiod = Quantity(row.rot_period, 'h')
            self.orbperiod = Quantity(row.orb_period, 'd')
            self.GM = Quantity(-float(row.mass) * const.G, 'm3/s2')
            self.moons = [SSObject(moon) for moon in
                          constants.loc[constants.orbits == self.object, 'Object'].to_list()]
            if len(self.moons) == 0:
                self.moons = None
            if self.orbits == 'Milky Way':
                self.type = 'Star'
                self.a = Quantity(row.a, 'km')
            elif self.orbits == 'Sun':
                self.type = 'Planet'
                self.a = Quantity(row.a, 'au')
            else:
                self.type = 'Moon'
                self.a = Quantity(row.a, 'km')
            register_unit('R_' + self.object, 'length', float(row.radius)*1e3)
        else:
            print(f'Object {obj} does not exist in table.')
            self.object = None

    def __len__(self):
        return 1 if self.moons is None else len(self.moons)+1

    def __eq__(self, other):
        return self.object == other.object

    def __hash__(self):
        return hash((self.object, ))

    def __repr__(self):
        return f'SSObject({self.object})'


def planet_dist(planet_, taa=None, time=None):
    """Distance from [au] and radial velocity relative to [km/s] the Sun at true anomaly taa.

    planet_dist.py:36-69: r = a(1-e^2)/(1+e cos nu); v_r from a finite difference of r over
    1001 mean-anomaly samples mapped to true anomaly by the 3-term equation-of-centre series,
    then np.interp at taa.
    """
    if isinstance(planet_, str):
        planet = SSObject(planet_)
        if planet.object is None:
            return None
    elif isinstance(planet_, SSObject):
        planet = planet_
    else:
        raise TypeError('solarsystemMB.planet_dist', 'Must give a SSObject or a object name.')

    if time is not None:
        raise NotImplementedError
    elif taa is not None:
        a = planet.a.value
        eps = planet.e
        taa_ = float(taa)
        if eps > 0:
            r = a * (1-eps**2)/(1+eps*np.cos(taa_))
            period = planet.orbperiod.value * 86400.
            time_ = np.linspace(0, 1, 1000)*period
            time_ = np.concatenate([np.array([time_[0]-time_[1]]), time_])
            mean_anomaly = np.linspace(0, 2*np.pi, 1000)
            mean_anomaly = np.concatenate(
                [np.array([mean_anomaly[0]-mean_anomaly[1]]), mean_anomaly])
            true_anomaly = (mean_anomaly +
                            (2*eps - eps**3/4)*np.sin(mean_anomaly) +
                            5/4 * eps**2 * np.sin(2*mean_anomaly) +
                            13/12 * eps**3 * np.sin(3*mean_anomaly))
            r_true = a * (1-eps**2)/(1+eps*np.cos(true_anomaly))
            drdt = (r_true[1:] - r_true[:-1])/(time_[1:] - time_[:-1])   # au/s
            v_r = np.interp(taa_, true_anomaly[1:], drdt * (const.AU_M/1e3))
            return Quantity(r, 'au'), Quantity(v_r, 'km/s')
        else:
            return Quantity(a, 'au'), Quantity(0., 'km/s')
    else:
        print('Neither a time nor a true anomaly was given.')
        return None
