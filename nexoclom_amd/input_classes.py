"""The seven specification classes an Input holds (parser side only).

Same names, attributes, defaults, clamping rules and error behaviour as the constructors in the
reference's initial_state/input_classes.py (Geometry :19-111, SurfaceInteraction :250-318,
Forces :419-431, SpatialDist :490-569, SpeedDist :702-761, AngularDist :905-960, Options
:1055-1100).  The PostgreSQL ``insert()``/``search()`` methods of the reference are out of scope
(SURVEY.md section 8b); the run catalogue lives in nexoclom_amd.Input instead.  Quantities are
nexoclom_amd.units.Quantity (float with .value) instead of astropy.
"""
import os

import numpy as np

from .solarsystem import SSObject
from .units import Quantity


class InputError(Exception):
    """Raised when a required parameter is not included (utilities/exceptions.py:2-6)."""

    def __init__(self, expression, message):
        self.expression = expression
        self.message = message
        super().__init__(expression, message)


class _Spec:
    _prefix = ''

    def __eq__(self, other):
        if not isinstance(other, type(self)):
            return False
        if set(self.__dict__) != set(other.__dict__):
            return False
        return all(self.__dict__[k] == other.__dict__[k] for k in self.__dict__)

    def __str__(self):
        return '\n'.join(f'{self._prefix}.{k} = {v}' for k, v in self.__dict__.items())

    __repr__ = __str__


def _rad(v):
    return Quantity(v, 'rad')


class Geometry(_Spec):
    _prefix = 'geometry'

    def __init__(self, gparam):
        planet = gparam.get('planet', None)
        if planet is None:
            raise InputError('Geometry.__init__', 'Planet not defined in inputfile.')
        self.planet = SSObject(planet.title())

        objlist = [self.planet.object]
        if self.planet.moons is not None:
            objlist.extend([m.object for m in self.planet.moons])

        self.startpoint = gparam.get('startpoint', self.planet.object).title()
        if self.startpoint not in objlist:
            print(f'{self.startpoint} is not a valid starting point.')
            olist = '\n\t'.join(objlist)
            print(f'Valid choices are:\n\t{olist}')
            raise ValueError

        if 'objects' in gparam:
            inc = set(i.strip().title() for i in gparam['objects'].split(','))
        else:
            inc = {self.planet.object, self.startpoint}
        for i in inc:
            if i not in objlist:
                raise InputError('Geometry.__init__', f'Invalid object {i} in geometry.include')
        self.objects = set(SSObject(o) for o in inc)
        if len(self.objects) == 0:
            self.objects = None

        if 'starttime' in gparam:
            self.type = 'geometry with starttime'
            # The reference parses an astropy Time here; only the string is kept (SPICE-based
            # geometry is out of scope and Output refuses this type, Output.py:95-96).
            self.time = gparam['starttime'].upper()
        else:
            self.type = 'geometry without starttime'
            if len(self.planet) == 1:
                self.phi = None
            elif 'phi' in gparam:
                phi = tuple(_rad(float(p)) for p in gparam['phi'].split(','))
                nmoons = len(self.objects - {self.planet})
                if len(phi) == nmoons:
                    self.phi = phi
                else:
                    raise InputError('Geometry.__init__',
                                     'The wrong number of orbital positions was given.')
            else:
                raise InputError('Geometry.__init__', 'geometry.phi was not specified.')

            if 'subsolarpoint' in gparam:
                subs = gparam['subsolarpoint'].split(',')
                try:
                    self.subsolarpoint = (_rad(float(subs[0])), _rad(float(subs[1])))
                except Exception:
                    raise InputError('Geometry.__init__',
                                     'The format for geometry.subsolarpoint is wrong.')
            else:
                self.subsolarpoint = (_rad(0), _rad(0))
            self.taa = _rad(float(gparam.get('taa', 0.)))


class SurfaceInteraction(_Spec):
    _prefix = 'surfaceinteraction'

    def __init__(self, sparam):
        sticktype = sparam['sticktype'].lower() if 'sticktype' in sparam else None
        if sticktype == 'temperature dependent':
            self.sticktype = sticktype
            if 'accomfactor' in sparam:
                self.accomfactor = float(sparam['accomfactor'])
            else:
                raise InputError('SurfaceInteraction.__init__',
                                 'surfaceinteraction.accomfactor not given.')
            if 'a' in sparam:
                A = tuple(float(a) for a in sparam['a'].split(','))
                if len(A) == 3:
                    self.A = A
                else:
                    raise InputError('SurfaceInteraction.__init__',
                                     'surfaceinteraction.A must have 3 values')
            else:
                self.A = (1.57014, -0.006262, 0.1614157)
        elif sticktype == 'surface map':
            self.sticktype = sticktype
            self.stick_mapfile = sparam.get('stick_mapfile', 'default')
            if not os.path.exists(self.stick_mapfile):
                print('Warning: stick_mapfile does not exist')
            self.stick_map = None
            self.subsolarlon = sparam.get('subsolarlon', None)
            if self.subsolarlon is not None:
                self.subsolarlon = _rad(float(self.subsolarlon))
            if 'accomfactor' in sparam:
                self.accomfactor = float(sparam['accomfactor'])
            else:
                raise InputError('SurfaceInteraction.__init__',
                                 'surfaceinteraction.accomfactor not given.')
        elif 'stickcoef' in sparam:
            self.sticktype = 'constant'
            self.stickcoef = min(max(float(sparam['stickcoef']), 0), 1)
            if 'accomfactor' in sparam:
                self.accomfactor = float(sparam['accomfactor'])
            elif self.stickcoef == 1:
                self.accomfactor = None
            else:
                raise InputError('SurfaceInteraction.__init__',
                                 'surfaceinteraction.accomfactor not given.')
        else:
            self.sticktype = 'constant'
            self.stickcoef = 1.
            self.accomfactor = None


def _parse_bool(text):
    t = text.strip().title()
    if t in ('True', '1'):
        return True
    if t in ('False', '0'):
        return False
    raise InputError('Forces.__init__', f'cannot interpret {text!r} as a boolean')


class Forces(_Spec):
    _prefix = 'forces'

    def __init__(self, fparam):
        self.gravity = _parse_bool(fparam['gravity']) if 'gravity' in fparam else True
        self.radpres = _parse_bool(fparam['radpres']) if 'radpres' in fparam else True


def _clamp(v, lo, hi):
    return min(max(v, lo), hi)


class SpatialDist(_Spec):
    _prefix = 'spatialdist'

    def __init__(self, sparam):
        if 'type' in sparam:
            self.type = sparam['type']
        else:
            raise InputError('SpatialDist.__init__', 'SpatialDist.type not given')

        if self.type == 'uniform':
            self.exobase = float(sparam['exobase']) if 'exobase' in sparam else 1.
            if 'longitude' in sparam:
                lon0, lon1 = (float(v.strip()) for v in sparam['longitude'].split(','))
                self.longitude = (_rad(_clamp(lon0, 0., 2*np.pi)), _rad(_clamp(lon1, 0., 2*np.pi)))
            else:
                self.longitude = (_rad(0.), _rad(2*np.pi))
            if 'latitude' in sparam:
                lat0, lat1 = (float(v.strip()) for v in sparam['latitude'].split(','))
                lat0 = _clamp(lat0, -np.pi/2, np.pi/2)
                lat1 = _clamp(lat1, -np.pi/2, np.pi/2)
                if lat0 > lat1:
                    raise InputError('SpatialDist.__init__',
                                     'SpatialDist.latitude[0] > SpatialDist.latitude[1]')
                self.latitude = (_rad(lat0), _rad(lat1))
            else:
                self.latitude = (_rad(-np.pi/2), _rad(np.pi/2))
        elif self.type == 'surface map':
            self.exobase = float(sparam['exobase']) if 'exobase' in sparam else 1.
            self.mapfile = sparam.get('mapfile', 'default')
            self.subsolarlon = sparam.get('subsolarlon', None)
            if self.subsolarlon is not None:
                self.subsolarlon = _rad(float(self.subsolarlon))
            self.coordinate_system = sparam.get('coordinate_system', 'solar-fixed')
        elif self.type == 'surface spot':
            self.exobase = float(sparam['exobase']) if 'exobase' in sparam else 1.
            for key in ('longitude', 'latitude', 'sigma'):
                if key in sparam:
                    setattr(self, key, _rad(float(sparam[key])))
                else:
                    raise InputError('SpatialDist.__init__', f'SpatialDist.{key} not given.')
        elif self.type == 'fitted output':
            self.unfit_outid = -1
            self.query = None
        else:
            raise InputError('SpatialDist.__init__',
                             f'SpatialDist.type = {self.type} not defined.')


class SpeedDist(_Spec):
    _prefix = 'speeddist'

    def __init__(self, sparam):
        self.type = sparam['type']

        def need(key, unit, attr=None):
            if key in sparam:
                setattr(self, attr or key, Quantity(float(sparam[key]), unit))
            else:
                raise InputError('SpatialDist.__init__', f'SpeedDist.{attr or key} not given.')

        if self.type == 'gaussian':
            need('vprob', 'km/s')
            need('sigma', 'km/s')
        elif self.type == 'sputtering':
            for key in ('alpha', 'beta'):
                if key in sparam:
                    setattr(self, key, float(sparam[key]))
                else:
                    raise InputError('SpatialDist.__init__', f'SpeedDist.{key} not given.')
            need('u', 'eV', 'U')
        elif self.type == 'maxwellian':
            need('temperature', 'K')
        elif self.type == 'flat':
            need('vprob', 'km/s')
            need('delv', 'km/s')
        elif self.type == 'user defined':
            self.vdistfile = sparam.get('vdistfile', 'default')
        elif self.type == 'fitted output':
            self.unfit_outid = -1
            self.query = None
        else:
            assert 0, f'SpeedDist.type = {self.type} not available'


class AngularDist(_Spec):
    _prefix = 'angulardist'

    def __init__(self, aparam):
        if 'type' in aparam:
            self.type = aparam['type'].lower()
            if self.type == 'radial':
                pass
            elif self.type == 'isotropic':
                if 'azimuth' in aparam:
                    az0, az1 = (float(v.strip()) for v in aparam['azimuth'].split(','))
                    self.azimuth = (_rad(_clamp(az0, 0., 2*np.pi)), _rad(_clamp(az1, 0., 2*np.pi)))
                else:
                    self.azimuth = (_rad(0), _rad(2*np.pi))
                self._altitude(aparam, np.pi/2)
            elif self.type == '2d':
                self._altitude(aparam, np.pi)
            else:
                raise InputError('AngularDist.__init__',
                                 f'AngularDist.type = {self.type} not defined.')
        else:
            self.type = 'isotropic'
            self.azimuth = (_rad(0), _rad(2*np.pi))
            self.altitude = (_rad(0), _rad(np.pi/2))

    def _altitude(self, aparam, top):
        if 'altitude' in aparam:
            alt0, alt1 = (float(v.strip()) for v in aparam['altitude'].split(','))
            alt0, alt1 = _clamp(alt0, 0, top), _clamp(alt1, 0, top)
            if alt0 > alt1:
                raise InputError('AngularDist.__init__',
                                 'AngularDist.altitude[0] > AngularDist.altitude[1]')
            self.altitude = (_rad(alt0), _rad(alt1))
        else:
            self.altitude = (_rad(0), _rad(top))


class Options(_Spec):
    _prefix = 'options'

    def __init__(self, oparam):
        if 'endtime' in oparam:
            self.endtime = Quantity(float(oparam['endtime']), 's')
        else:
            raise InputError('Options.__init__', 'options.endtime not specified.')

        if 'species' in oparam:
            self.species = oparam['species'].capitalize()
        elif 'atom' in oparam:
            self.species = oparam['atom'].capitalize()
        else:
            raise InputError('Options.__init__', 'options.species not specified.')

        self.lifetime = Quantity(float(oparam.get('lifetime', 0)), 's')

        if 'outeredge' in oparam:
            self.outeredge = float(oparam['outeredge'])
        elif 'outer_edge' in oparam:
            self.outeredge = float(oparam['outer_edge'])
        else:
            self.outeredge = 1e30

        # The reference reads oparam['step_size'] in its 'stepsize' branch (a KeyError,
        # input_classes.py:1086-1087); the evident intent is honoured here.
        if 'step_size' in oparam:
            self.step_size = float(oparam['step_size'])
        elif 'stepsize' in oparam:
            self.step_size = float(oparam['stepsize'])
        else:
            self.step_size = 0.

        if self.step_size == 0:
            # The reference leaves a file-supplied resolution as a STRING
            # (input_classes.py:1092), which the variable-step driver cannot compare; it is
            # converted here.
            self.resolution = float(oparam.get('resolution', 1e-4))
        else:
            self.resolution = None

        if 'fitted' in oparam:
            self.fitted = oparam['fitted'].casefold() == 'True'.casefold()
        else:
            self.fitted = False

        # EXTENSION (no counterpart in the reference, whose charge-exchange term is a
        # commented stub, state.py:56-70): loss in a plasma torus around the planet,
        #   rate = chx_rate exp(-((rho - chx_rho0)/chx_width)^2 - (z/chx_height)^2)
        #          [* |v - v_corotation| / v_corotation(chx_rho0)  if chx_corotation]
        # lengths in planet radii, chx_rate in 1/s.  Absent chx_rate = no such loss.
        if 'chx_rate' in oparam:
            self.chx = {'k0': float(oparam['chx_rate']),
                        'rho0': float(oparam.get('chx_rho0', 5.9)),
                        'width': float(oparam.get('chx_width', 1.0)),
                        'height': float(oparam.get('chx_height', 1.0)),
                        'corotation': oparam.get('chx_corotation', 'false').casefold() == 'true'}
        else:
            self.chx = None
