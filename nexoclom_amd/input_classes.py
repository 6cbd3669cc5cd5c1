"""The seven sections of an inputfile as declarative specifications.

The reference builds each section object with a hand-written constructor
(initial_state/input_classes.py: Geometry :19-111, SurfaceInteraction :250-318, Forces :419-431,
SpatialDist :490-569, SpeedDist :702-761, AngularDist :905-960, Options :1055-1100).  Here a
section is DATA: a tuple of ``Field`` records -- attribute name, accepted keys, a text -> value
converter, a default or the error raised when the key is missing -- optionally switched by the
section's ``type`` key, and one generic ``Section.__init__`` walks it.  What is kept from the
reference is the contract a drop-in needs: class names, attribute names and their order in
``__dict__``, defaults, clamping rules, which omissions raise ``InputError`` (and with which
text), equality and the ``section.key = value`` rendering.

Not here: the PostgreSQL ``insert()`` / ``search()`` methods (out of scope, SURVEY.md section 8b;
the run catalogue lives on nexoclom_amd.Input) and astropy -- quantities are
nexoclom_amd.units.Quantity (a float that answers ``.value`` / ``.unit``).
"""
import math
import os

from .solarsystem import SSObject
from .units import Quantity

TWO_PI = 2*math.pi
HALF_PI = math.pi/2


class InputError(Exception):
    """A required parameter is missing or malformed (utilities/exceptions.py:2-6)."""

    def __init__(self, expression, message):
        self.expression = expression
        self.message = message
        super().__init__(expression, message)


# ---- converters: the text right of '=' -> attribute value ---------------------------------------
def number(text):
    return float(text)


def measured(unit):
    """float with a unit label."""
    return lambda text: Quantity(float(text), unit)


def numbers(text):
    return tuple(float(part) for part in text.split(','))


def limited(lo, hi):
    """A number pushed into [lo, hi]."""
    return lambda text: min(max(float(text), lo), hi)


def angle_pair(lo, hi):
    """Two comma-separated angles, each pushed into [lo, hi], as radian quantities."""
    def convert(text):
        first, second = (min(max(float(part.strip()), lo), hi) for part in text.split(','))
        return Quantity(first, 'rad'), Quantity(second, 'rad')
    return convert


def radians(*values):
    return tuple(Quantity(v, 'rad') for v in values)


def switch(text):
    """'True' / 'False' / '1' / '0' in any case (the reference eval()s the title-cased text)."""
    word = text.strip().casefold()
    if word in ('true', '1'):
        return True
    if word in ('false', '0'):
        return False
    raise InputError('Forces.__init__', f'cannot interpret {text!r} as a boolean')


_REQUIRED = object()


class Field:
    """One attribute of a section.

    attr     name of the attribute set on the section object
    convert  text -> value
    default  value used when none of ``keys`` is present (``missing`` text -> InputError instead)
    keys     accepted inputfile keys, first match wins (default: the attribute name, case-folded)
    """
    __slots__ = ('attr', 'convert', 'default', 'keys', 'missing', 'blame')

    def __init__(self, attr, convert=str, default=_REQUIRED, keys=None, missing=None, blame=None):
        self.attr = attr
        self.convert = convert
        self.default = default
        self.keys = tuple(keys) if keys else (attr.casefold(),)
        self.missing = missing
        self.blame = blame

    def value(self, params, section):
        for key in self.keys:
            if key in params:
                return self.convert(params[key])
        if self.default is _REQUIRED:
            raise InputError(f'{self.blame or type(section).__name__}.__init__',
                             self.missing or f'{type(section).__name__}.{self.attr} not given.')
        return self.default


class Constant:
    """An attribute that does not come from the file."""
    __slots__ = ('attr', 'fixed')

    def __init__(self, attr, fixed):
        self.attr, self.fixed = attr, fixed

    def value(self, params, section):
        return self.fixed


class Section:
    """Generic section: ``layout(params)`` names the fields, ``check()`` validates across them."""
    prefix = ''
    fields = ()

    def __init__(self, params):
        for item in self.layout(params):
            setattr(self, item.attr, item.value(params, self))
        self.check(params)

    def layout(self, params):
        return self.fields

    def check(self, params):
        pass

    def __eq__(self, other):
        return (isinstance(other, type(self)) and self.__dict__.keys() == other.__dict__.keys()
                and all(other.__dict__[k] == v for k, v in self.__dict__.items()))

    def __ne__(self, other):
        return not self == other

    __hash__ = None

    def __str__(self):
        return '\n'.join(f'{self.prefix}.{k} = {v}' for k, v in self.__dict__.items())

    __repr__ = __str__


# ---- geometry -----------------------------------------------------------------------------------
class Geometry(Section):
    """Which bodies take part and where they are.  The body bookkeeping (planet, its moons, the
    start point, the included set) is resolved first; the remaining attributes are fields."""
    prefix = 'geometry'

    def layout(self, params):
        if 'planet' not in params:
            raise InputError('Geometry.__init__', 'Planet not defined in inputfile.')
        planet = SSObject(params['planet'].title())
        family = [planet.object] + [m.object for m in (planet.moons or ())]
        start = params.get('startpoint', planet.object).title()
        if start not in family:
            print(f'{start} is not a valid starting point.')
            print('Valid choices are:\n\t' + '\n\t'.join(family))
            raise ValueError
        if 'objects' in params:
            wanted = {name.strip().title() for name in params['objects'].split(',')}
        else:
            wanted = {planet.object, start}
        for name in wanted:
            if name not in family:
                raise InputError('Geometry.__init__', f'Invalid object {name} in geometry.include')
        bodies = {SSObject(name) for name in wanted} or None
        head = [Constant('planet', planet), Constant('startpoint', start),
                Constant('objects', bodies)]
        if 'starttime' in params:
            # The reference parses an astropy Time here; only the text is kept: SPICE-driven
            # geometry is out of scope and Output refuses this type (Output.py:95-96).
            return head + [Constant('type', 'geometry with starttime'),
                           Constant('time', params['starttime'].upper())]
        moons_included = len((bodies or set()) - {planet})
        return head + [Constant('type', 'geometry without starttime'),
                       Constant('phi', self._orbital_phases(params, planet, moons_included)),
                       Field('subsolarpoint', self._subsolar, default=radians(0, 0)),
                       Field('taa', measured('rad'), default=Quantity(0., 'rad'))]

    @staticmethod
    def _orbital_phases(params, planet, expected):
        if len(planet) == 1:
            return None                      # a planet without moons has nothing to place
        if 'phi' not in params:
            raise InputError('Geometry.__init__', 'geometry.phi was not specified.')
        phases = radians(*numbers(params['phi']))
        if len(phases) != expected:
            raise InputError('Geometry.__init__',
                             'The wrong number of orbital positions was given.')
        return phases

    @staticmethod
    def _subsolar(text):
        try:
            lon, lat = text.split(',')[:2]
            return radians(float(lon), float(lat))
        except (ValueError, TypeError):
            raise InputError('Geometry.__init__',
                             'The format for geometry.subsolarpoint is wrong.') from None


# ---- surface interaction --------------------------------------------------------------------------
_ACCOM = Field('accomfactor', number, blame='SurfaceInteraction',
               missing='surfaceinteraction.accomfactor not given.')


def _three_coefficients(text):
    coef = numbers(text)
    if len(coef) != 3:
        raise InputError('SurfaceInteraction.__init__', 'surfaceinteraction.A must have 3 values')
    return coef


def _optional_angle(text):
    return Quantity(float(text), 'rad')


class SurfaceInteraction(Section):
    prefix = 'surfaceinteraction'
    by_sticktype = {
        'temperature dependent': (
            Constant('sticktype', 'temperature dependent'), _ACCOM,
            Field('A', _three_coefficients, default=(1.57014, -0.006262, 0.1614157))),
        'surface map': (
            Constant('sticktype', 'surface map'),
            Field('stick_mapfile', default='default'),
            Constant('stick_map', None),       # the reference's pickled SourceMap: out of scope
            Field('subsolarlon', _optional_angle, default=None), _ACCOM),
    }

    def layout(self, params):
        kind = params['sticktype'].lower() if 'sticktype' in params else None
        if kind in self.by_sticktype:
            return self.by_sticktype[kind]
        if 'stickcoef' not in params:          # nothing said: packets stick where they land
            return (Constant('sticktype', 'constant'), Constant('stickcoef', 1.),
                    Constant('accomfactor', None))
        stick = limited(0, 1)(params['stickcoef'])
        # a perfectly sticking surface needs no accommodation factor; any other must give one
        accom = Field('accomfactor', number, default=None) if stick == 1 else _ACCOM
        return (Constant('sticktype', 'constant'), Constant('stickcoef', stick), accom)

    def check(self, params):
        if self.sticktype == 'surface map' and not os.path.exists(self.stick_mapfile):
            print('Warning: stick_mapfile does not exist')


# ---- forces -----------------------------------------------------------------------------------------
class Forces(Section):
    prefix = 'forces'
    fields = (Field('gravity', switch, default=True), Field('radpres', switch, default=True))


# ---- spatial distribution -----------------------------------------------------------------------
_EXOBASE = Field('exobase', number, default=1.)


def _spot(attr):
    return Field(attr, measured('rad'), blame='SpatialDist',
                 missing=f'SpatialDist.{attr} not given.')


class SpatialDist(Section):
    prefix = 'spatialdist'
    by_type = {
        'uniform': (
            _EXOBASE,
            Field('longitude', angle_pair(0., TWO_PI), default=radians(0., TWO_PI)),
            Field('latitude', angle_pair(-HALF_PI, HALF_PI), default=radians(-HALF_PI, HALF_PI))),
        'surface map': (
            _EXOBASE, Field('mapfile', default='default'),
            Field('subsolarlon', _optional_angle, default=None),
            Field('coordinate_system', default='solar-fixed')),
        'surface spot': (_EXOBASE, _spot('longitude'), _spot('latitude'), _spot('sigma')),
        'fitted output': (Constant('unfit_outid', -1), Constant('query', None)),
    }

    def layout(self, params):
        if 'type' not in params:
            raise InputError('SpatialDist.__init__', 'SpatialDist.type not given')
        kind = params['type']
        if kind not in self.by_type:
            raise InputError('SpatialDist.__init__', f'SpatialDist.type = {kind} not defined.')
        return (Constant('type', kind),) + self.by_type[kind]

    def check(self, params):
        if self.type == 'uniform' and self.latitude[0] > self.latitude[1]:
            raise InputError('SpatialDist.__init__',
                             'SpatialDist.latitude[0] > SpatialDist.latitude[1]')


# ---- speed distribution ---------------------------------------------------------------------------
def _speed(attr, convert, key=None):
    # the reference blames SpatialDist for a missing SpeedDist parameter; kept, it is part of what
    # callers can observe
    return Field(attr, convert, keys=(key or attr.casefold(),), blame='SpatialDist',
                 missing=f'SpeedDist.{attr} not given.')


class SpeedDist(Section):
    prefix = 'speeddist'
    by_type = {
        'gaussian': (_speed('vprob', measured('km/s')), _speed('sigma', measured('km/s'))),
        'sputtering': (_speed('alpha', number), _speed('beta', number),
                       _speed('U', measured('eV'), key='u')),
        'maxwellian': (_speed('temperature', measured('K')),),
        'flat': (_speed('vprob', measured('km/s')), _speed('delv', measured('km/s'))),
        'user defined': (Field('vdistfile', default='default'),),
        'fitted output': (Constant('unfit_outid', -1), Constant('query', None)),
    }

    def layout(self, params):
        kind = params['type']                   # KeyError when absent, as in the reference
        assert kind in self.by_type, f'SpeedDist.type = {kind} not available'
        return (Constant('type', kind),) + self.by_type[kind]


# ---- angular distribution -----------------------------------------------------------------------
def _altitude(top):
    return Field('altitude', angle_pair(0, top), default=radians(0, top))


class AngularDist(Section):
    prefix = 'angulardist'
    by_type = {
        'radial': (),
        'isotropic': (Field('azimuth', angle_pair(0., TWO_PI), default=radians(0, TWO_PI)),
                      _altitude(HALF_PI)),
        '2d': (_altitude(math.pi),),
    }

    def layout(self, params):
        if 'type' not in params:                # nothing said: the isotropic defaults, untouched
            return (Constant('type', 'isotropic'), Constant('azimuth', radians(0, TWO_PI)),
                    Constant('altitude', radians(0, HALF_PI)))
        kind = params['type'].lower()
        if kind not in self.by_type:
            raise InputError('AngularDist.__init__', f'AngularDist.type = {kind} not defined.')
        return (Constant('type', kind),) + self.by_type[kind]

    def check(self, params):
        alt = getattr(self, 'altitude', None)
        if alt is not None and alt[0] > alt[1]:
            raise InputError('AngularDist.__init__',
                             'AngularDist.altitude[0] > AngularDist.altitude[1]')


# ---- options ------------------------------------------------------------------------------------------
def _torus(params):
    """EXTENSION (the reference's charge-exchange term is a commented stub, state.py:56-70): loss
    in a plasma torus around the planet,
        rate = chx_rate exp(-((rho - chx_rho0)/chx_width)^2 - (z/chx_height)^2)
               [* |v - v_corotation| / v_corotation(chx_rho0)  if chx_corotation]
    lengths in planet radii, chx_rate in 1/s.  No chx_rate: no such loss."""
    if 'chx_rate' not in params:
        return None
    return {'k0': float(params['chx_rate']), 'rho0': float(params.get('chx_rho0', 5.9)),
            'width': float(params.get('chx_width', 1.0)),
            'height': float(params.get('chx_height', 1.0)),
            'corotation': params.get('chx_corotation', 'false').casefold() == 'true'}


class Options(Section):
    prefix = 'options'

    def layout(self, params):
        # 'stepsize' is honoured (the reference looks up the wrong key in that branch and raises
        # KeyError, input_classes.py:1086-1087); a file-supplied resolution becomes a float (the
        # reference keeps the text, :1092, which its variable-step driver cannot compare).
        step = Field('step_size', number, default=0., keys=('step_size', 'stepsize'))
        adaptive = step.value(params, self) == 0
        return (
            Field('endtime', measured('s'), missing='options.endtime not specified.'),
            Field('species', str.capitalize, keys=('species', 'atom'),
                  missing='options.species not specified.'),
            Field('lifetime', measured('s'), default=Quantity(0., 's')),
            Field('outeredge', number, default=1e30, keys=('outeredge', 'outer_edge')),
            step,
            Field('resolution', number, default=1e-4) if adaptive else Constant('resolution', None),
            Field('fitted', lambda text: text.casefold() == 'true', default=False),
            Constant('chx', _torus(params)),
        )
