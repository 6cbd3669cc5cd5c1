"""g-values, radiation acceleration and photo-loss rates (host side; tables feed the kernels).

Re-statement without astropy/periodictable of the reference's
  atomicdata/g_values.py:59-94      gValue        g(v) for one line, scaled (r_ref/a)^2, sorted by v
  atomicdata/g_values.py:134-160    RadPresConst  a_rad(v) = sum_lines h/(m lambda) g_line(v)
  atomicdata/photolossrates.py:66-86 PhotoRate    sum of kappa/a^2 over ALL reactions of a species
  atomicdata/atomicmass.py:5-51     atomicmass
The tables come from nexoclom_amd/data/*.csv (built by tools/make_data.py from the reference's
text data files).
"""
import functools
import os

import numpy as np
import pandas as pd

from . import constants as const
from .units import Quantity

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


@functools.lru_cache(maxsize=None)
def _gvalue_table():
    return pd.read_csv(os.path.join(_DATA, 'gvalues.csv'))


@functools.lru_cache(maxsize=None)
def _photo_table():
    return pd.read_csv(os.path.join(_DATA, 'photorates.csv'))


def atomicmass(species):
    """Atomic mass in u (atomicdata/atomicmass.py:5-51); None when unknown."""
    if species in const.ATOMIC_MASS:
        return Quantity(const.ATOMIC_MASS[species], 'u')
    print(f'WARNING: mathMB.atomicmass: {species} not found')
    return None


def _au(aplanet):
    return float(aplanet)


class gValue:
    """g-value vs. radial velocity for (species, wavelength) at heliocentric distance aplanet.

    ``velocity`` [km/s] ascending, ``g`` [1/s] (atomicdata/g_values.py:75-91).  Unknown
    species/line gives the reference's two-point zero table (:78-83).
    """

    def __init__(self, sp, wavelength, aplanet=1.0):
        self.species = sp
        self.wavelength = Quantity(float(wavelength), 'AA')
        self.aplanet = Quantity(_au(aplanet), 'au')
        tab = _gvalue_table()
        gvalue = tab[(tab.species == sp) & (tab.wavelength == float(wavelength))]
        if len(gvalue) == 0:
            self.velocity = np.array([0., 1.])
            self.g = np.array([0., 0.])
            self.filename = None
            print(f'Warning: g-values not found for species = {sp}')
        elif len(gvalue.source.unique()) == 1:
            velocity = gvalue.velocity.values.astype(float)
            g = (gvalue.gvalue * gvalue.refpoint**2 / self.aplanet.value**2).values
            s = np.argsort(velocity)
            self.velocity, self.g = velocity[s], g[s]
            self.filename = gvalue.source.unique()[0]
        else:
            print('This should never happen')
            raise ValueError()


class RadPresConst:
    """Radiation acceleration vs. radial velocity (atomicdata/g_values.py:134-160).

    ``velocity`` [km/s] = sorted union of all the species' g-value velocity grids (:146);
    ``accel`` [km/s^2] = sum over the species' lines of h/(m lambda) * interp(g_line) (:150-154).
    """

    def __init__(self, species, aplanet):
        self.species = species
        self.aplanet = Quantity(_au(aplanet), 'au')
        tab = _gvalue_table()
        if species in tab.species.values:
            subset = tab.loc[tab.species == species]
            self.wavelength = np.array(sorted(subset.wavelength.unique()))
            self.velocity = np.array(sorted(subset.velocity.unique()))
            rpres = np.zeros_like(self.velocity)
            mass_kg = atomicmass(species).value * const.AMU
            for wave in self.wavelength:
                gval = gValue(species, wave, aplanet)
                g_ = np.interp(self.velocity, gval.velocity, gval.g)
                # h / m / lambda * g  [m/s^2] -> km/s^2
                rpres_ = const.H_PLANCK / mass_kg / (wave * 1e-10) * g_
                rpres += rpres_ * 1e-3
            self.accel = rpres
        else:
            self.velocity = np.array([0., 1.])
            self.accel = np.array([0., 0.])
            print(f'Warning: g-values not found for species = {species}')


class PhotoRate:
    """Total photo-loss rate [1/s] of a species at aplanet [au] (photolossrates.py:66-86)."""

    def __init__(self, species, aplanet_=1.0):
        tab = _photo_table()
        prates = tab[tab.species == species]
        aplanet = _au(aplanet_)
        self.species = species
        self.aplanet = Quantity(aplanet, 'au')
        if len(prates) == 0:
            print('No photoreactions found')
            self.reactions = None
            self.rate = Quantity(1e-30, '1/s')
        else:
            rates = prates['kappa'].apply(lambda k: k/aplanet**2).values
            self.reactions = prates
            self.rate = Quantity(rates.sum(), '1/s')

    def __str__(self):
        return (f'Species = {self.species}\nDistance = {self.aplanet}\nRate = {self.rate}')


class LossInfo:
    """Loss processes for a run (initial_state/LossInfo.py:5-35): photo = |1/lifetime| for a
    negative lifetime, PhotoRate(species, aplanet) for lifetime == 0."""

    def __init__(self, atom, lifetime, aplanet):
        self.photo = 0.
        self.eimp = 0.
        self.chX = 0.
        self.reactions = []
        lifetime_ = float(lifetime)
        if lifetime_ < 0:
            self.photo = np.abs(1./lifetime_)
            self.reactions = 'Generic photo reaction'
        elif lifetime_ == 0:
            photo = PhotoRate(atom, aplanet)
            self.photo = photo.rate.value
            self.reactions = (photo.reactions['reaction'].values
                              if photo.reactions is not None else [])
        else:
            print('LossInfo objects should not be instantiated with lifetime > 0')
        if len(self.reactions) == 0:
            self.reactions = None

    def __len__(self):
        return len(self.reactions) if self.reactions is not None else 0
