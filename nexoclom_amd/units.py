"""Minimal stand-in for the astropy Quantities the reference exposes on its objects.

The reference stores e.g. ``inputs.options.endtime`` as ``float*u.s`` and reads it back with
``.value`` (particle_tracking/Output.py:113,375; particle_tracking/state.py:46).  astropy is not
a dependency here, so a Quantity is a ``float`` that also answers ``.value``, ``.unit`` and
``.to(unit)`` for the handful of conversions the hot path needs.  Arithmetic on it returns plain
floats, which is what every consumer on this path does with ``.value`` anyway.
"""
import math

# unit -> (dimension, scale to SI)
_SCALE = {
    's': ('time', 1.0), 'h': ('time', 3600.0), 'd': ('time', 86400.0),
    'm': ('length', 1.0), 'cm': ('length', 1e-2), 'km': ('length', 1e3),
    'au': ('length', 1.495978707e11),
    'rad': ('angle', 1.0), 'deg': ('angle', math.pi/180.0),
    'km/s': ('speed', 1e3), 'm/s': ('speed', 1.0), 'cm/s': ('speed', 1e-2),
    'km/s2': ('accel', 1e3), 'cm/s2': ('accel', 1e-2), 'm/s2': ('accel', 1.0),
    '1/s': ('rate', 1.0), 'K': ('temperature', 1.0), 'eV': ('energy', 1.602176634e-19),
    'kg': ('mass', 1.0), 'u': ('mass', 1.66053906660e-27), 'AA': ('length', 1e-10),
    'm3/s2': ('gm', 1.0), 'cm2': ('area', 1e-4), 'km2': ('area', 1e6),
}


def register_unit(name, dimension, scale):
    """Register a derived unit, e.g. ``R_Mercury`` (length, 2440530 m)."""
    _SCALE[name] = (dimension, float(scale))


class Quantity(float):
    """float carrying a unit label; ``.value`` returns the bare float."""

    def __new__(cls, value, unit=''):
        obj = super().__new__(cls, float(value))
        obj.unit = unit
        return obj

    @property
    def value(self):
        return float(self)

    def to(self, unit):
        d0, s0 = _SCALE[self.unit]
        d1, s1 = _SCALE[unit]
        if d0 != d1:
            raise ValueError(f'cannot convert {self.unit} to {unit}')
        return Quantity(float(self)*s0/s1, unit)

    def __repr__(self):
        return f'{float(self)!r} {self.unit}'.strip()

    __str__ = __repr__

    def __reduce__(self):
        return (Quantity, (float(self), self.unit))

    def __eq__(self, other):
        return float(self) == float(other)

    def __hash__(self):
        return hash(float(self))
