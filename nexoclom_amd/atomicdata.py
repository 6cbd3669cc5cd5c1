"""g-values, radiation acceleration and photo-loss rates (host side; the tables feed the kernels).

Drop-in, without astropy / periodictable / pandas pickles, for the reference's
  atomicdata/g_values.py:59-94       gValue        g(v) of one line, scaled (r_ref/a)^2, sorted by v
  atomicdata/g_values.py:134-160     RadPresConst  a_rad(v) = sum over lines of h/(m lambda) g(v)
  atomicdata/photolossrates.py:66-86 PhotoRate     sum of kappa/a^2 over ALL reactions of a species
  atomicdata/atomicmass.py:5-51      atomicmass
  initial_state/LossInfo.py:5-35     LossInfo

Data: nexoclom_amd/data/gvalues.csv and photorates.csv, written by tools/make_data.py from the
reference's TEXT data files with 17 significant digits.  They are parsed here with Python's
correctly rounded ``float`` into one in-memory index per file -- {(species, wavelength): line} and
{species: reactions} -- built on first use, so the doubles are exactly the ones make_data wrote
and a lookup is a dict access, not a table scan.
"""
import csv
import functools
import os
from collections import namedtuple

import numpy as np

from . import constants as const
from .units import Quantity

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')

Line = namedtuple('Line', 'velocity gvalue refpoint sources')       # arrays in file order
Reaction = namedtuple('Reaction', 'reaction kappa reference')


@functools.lru_cache(maxsize=None)
def _lines():
    """{(species, wavelength): Line} plus, under the key species, its wavelengths."""
    rows = {}
    with open(os.path.join(_DATA, 'gvalues.csv'), newline='') as handle:
        for row in csv.DictReader(handle):
            key = (row['species'], float(row['wavelength']))
            rows.setdefault(key, []).append((float(row['velocity']), float(row['gvalue']),
                                             float(row['refpoint']), row['source']))
    index = {}
    for (species, wavelength), recs in rows.items():
        v, g, ref, src = zip(*recs)
        index[(species, wavelength)] = Line(np.array(v), np.array(g), np.array(ref),
                                            tuple(dict.fromkeys(src)))
        index.setdefault(species, []).append(wavelength)
    return index


@functools.lru_cache(maxsize=None)
def _reactions():
    table = {}
    with open(os.path.join(_DATA, 'photorates.csv'), newline='') as handle:
        for row in csv.DictReader(handle):
            table.setdefault(row['species'], []).append(
                Reaction(row['reaction'], float(row['kappa']), row['reference']))
    return table


def atomicmass(species):
    """Atomic mass in u (atomicdata/atomicmass.py:5-51); None when unknown."""
    mass = const.ATOMIC_MASS.get(species)
    if mass is None:
        print(f'WARNING: mathMB.atomicmass: {species} not found')
        return None
    return Quantity(mass, 'u')


def _flat_table():
    """The reference's answer for an unknown species or line: two points, all zero."""
    return np.array([0., 1.]), np.array([0., 0.])


class gValue:
    """g-value [1/s] vs. radial velocity [km/s] of one emission line at heliocentric distance
    ``aplanet`` [au]: the tabulated values times (reference distance / aplanet)^2, ascending in
    velocity (atomicdata/g_values.py:75-91)."""

    def __init__(self, sp, wavelength, aplanet=1.0):
        self.species, self.filename = sp, None
        self.wavelength = Quantity(float(wavelength), 'AA')
        self.aplanet = Quantity(float(aplanet), 'au')
        line = _lines().get((sp, float(wavelength)))
        if line is None:
            self.velocity, self.g = _flat_table()
            print(f'Warning: g-values not found for species = {sp}')
            return
        if len(line.sources) != 1:          # two files claim one line (g_values.py:79-81)
            raise ValueError('This should never happen')
        order = np.argsort(line.velocity)
        self.velocity = line.velocity[order]
        self.g = (line.gvalue * line.refpoint**2 / self.aplanet.value**2)[order]
        self.filename = line.sources[0]


class RadPresConst:
    """Radiation acceleration [km/s^2] vs. radial velocity [km/s] of a species
    (atomicdata/g_values.py:134-160): every line's g(v) is interpolated onto the sorted union of
    the species' velocity grids and contributes h / (m lambda) g."""

    def __init__(self, species, aplanet):
        self.species, self.aplanet = species, Quantity(float(aplanet), 'au')
        waves = _lines().get(species)
        if not waves:
            self.velocity, self.accel = _flat_table()
            print(f'Warning: g-values not found for species = {species}')
            return
        self.wavelength = np.array(sorted(waves))
        self.velocity = np.unique(np.concatenate([_lines()[(species, w)].velocity for w in waves]))
        mass_kg = atomicmass(species).value * const.AMU
        total = np.zeros_like(self.velocity)
        for wave in sorted(waves):
            line = gValue(species, wave, aplanet)
            on_grid = np.interp(self.velocity, line.velocity, line.g)
            momentum_kick = const.H_PLANCK / mass_kg / (wave * 1e-10) * on_grid      # m/s^2
            total += momentum_kick * 1e-3
        self.accel = total


class PhotoRate:
    """Total photo-loss rate [1/s] of a species at ``aplanet_`` [au]: every tabulated reaction of
    the species counts, each scaled 1/a^2 (photolossrates.py:66-86)."""

    def __init__(self, species, aplanet_=1.0):
        aplanet = float(aplanet_)
        self.species, self.aplanet = species, Quantity(aplanet, 'au')
        found = _reactions().get(species)
        self.reactions = found or None
        if not found:
            print('No photoreactions found')
            self.rate = Quantity(1e-30, '1/s')
            return
        self.rate = Quantity(np.array([r.kappa/aplanet**2 for r in found]).sum(), '1/s')

    def __str__(self):
        return f'Species = {self.species}\nDistance = {self.aplanet}\nRate = {self.rate}'


class LossInfo:
    """What removes atoms during a run (initial_state/LossInfo.py:5-35): a negative lifetime
    means a generic photo-process of rate 1/|lifetime|; lifetime 0 means the tabulated
    photo-reactions of the species at the planet's distance."""

    def __init__(self, atom, lifetime, aplanet):
        self.photo, self.eimp, self.chX, self.reactions = 0., 0., 0., None
        lifetime = float(lifetime)
        if lifetime > 0:
            print('LossInfo objects should not be instantiated with lifetime > 0')
        elif lifetime < 0:
            self.photo = abs(1./lifetime)
            self.reactions = 'Generic photo reaction'
        else:
            tabulated = PhotoRate(atom, aplanet)
            self.photo = tabulated.rate.value
            names = [r.reaction for r in (tabulated.reactions or ())]
            self.reactions = np.array(names) if names else None

    def __len__(self):
        return 0 if self.reactions is None else len(self.reactions)
