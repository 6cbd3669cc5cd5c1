"""Initial packet states X0 (host side, NumPy; draw-for-draw identical to the reference).

Re-statement of initial_state/source_distribution.py:12-283, math/randomdeviates.py:8-83 and
math/distributions.py:7-21 of the reference.  The order of draws from ``output.randgen`` is the
reference's: time (variable-step only) -> sin(lat) -> lon -> speed -> sin(alt) -> az, so a seeded
run produces the same X0 (SURVEY.md section 7 "Seed parity").  'maxwellian', 'sputtering' and
'surface spot' draw from the UNSEEDED global ``numpy.random`` exactly as the reference does
(randomdeviates.py:33,63-65).  Map-file driven sources ('surface map', 'user defined') are out
of scope.
"""
import numpy as np
import numpy.random as random

from . import constants as const
from .input_classes import InputError


def xyz_from_lonlat(lon, lat, isplan, exobase):
    """source_distribution.py:12-34.  Planet: lon 0 = subsolar point (0,-1,0), 90 deg = dusk
    (1,0,0).  Satellite: lon 0 = sub-planet point, 90 deg = leading point (-1,0,0)."""
    sign = 1.0 if isplan else -1.0
    x0 = sign * exobase * np.sin(lon) * np.cos(lat)
    y0 = -exobase * np.cos(lon) * np.cos(lat)
    z0 = exobase * np.sin(lat)
    X0 = np.array([x0, y0, z0])
    assert np.all(np.isfinite(X0)), 'Non-Finite values of X0'
    return X0


def random_deviates_1d(x, f_x, num):
    """Transformation-method deviates from f_x on x (randomdeviates.py:8-33; global RNG)."""
    x_ = np.linspace(x.min(), x.max(), f_x.shape[0])
    cumsum = f_x.cumsum()
    cumsum -= cumsum.min()
    cumsum /= cumsum.max()
    return np.interp(random.rand(num), cumsum, x_)


def random_deviates_2d(fdist, x0, y0, num):
    """Acceptance/rejection deviates from a 2-D map (randomdeviates.py:36-83; global RNG)."""
    from scipy import interpolate
    mx = (x0.max()-x0.min(), x0.min())
    my = (y0.max()-y0.min(), y0.min())
    fmax = fdist.max()
    x0_ = np.linspace(x0.min(), x0.max(), fdist.shape[0])
    y0_ = np.linspace(y0.min(), y0.max(), fdist.shape[1])
    xpts, ypts = [], []
    while len(xpts) < num:
        ux = random.rand(num)*mx[0] + mx[1]
        uy = random.rand(num)*my[0] + my[1]
        uf = random.rand(num)*fmax
        val = interpolate.interpn((x0_, y0_), fdist, (ux, uy))
        mm = uf < val
        xpts.extend(list(ux[mm]))
        ypts.extend(list(uy[mm]))
    return np.array(xpts[0:num]), np.array(ypts[0:num])


def _mass_kg(species):
    return const.ATOMIC_MASS[species] * const.AMU


def sputdist(velocity, U_eV, alpha, beta, species):
    """Sputtering speed distribution on velocity [km/s] (distributions.py:7-13)."""
    v_b = np.sqrt(2*U_eV*const.EV/_mass_kg(species)) / 1e3
    f_v = velocity**(2*beta+1) / (velocity**2 + v_b**2)**alpha
    return f_v / np.max(f_v)


def MaxwellianDist(velocity, temperature, species):
    """Maxwellian flux distribution on velocity [km/s] (distributions.py:16-21)."""
    vth2 = 2*temperature*const.K_B/_mass_kg(species) / 1e6
    f_v = velocity**3 * np.exp(-velocity**2/vth2)
    return f_v / np.max(f_v)


def surface_distribution(outputs):
    """Launch positions on the sphere r = exobase (source_distribution.py:37-134)."""
    spatialdist = outputs.inputs.spatialdist
    npack = outputs.npackets

    if spatialdist.type == 'uniform':
        ll = tuple(map(np.sin, spatialdist.latitude))
        sinlat = ll[0] + (ll[1]-ll[0]) * outputs.randgen.random(npack)
        lat = np.arcsin(sinlat)
        lon0 = [float(v) for v in spatialdist.longitude]
        if lon0[0] > lon0[1]:
            lon0 = [lon0[0], lon0[1]+2*np.pi]
        lon = (lon0[0] + (lon0[1]-lon0[0]) * outputs.randgen.random(npack)) % (2*np.pi)
    elif spatialdist.type == 'surface spot':
        lon0, lat0, sigma0 = (float(spatialdist.longitude), float(spatialdist.latitude),
                              float(spatialdist.sigma))
        spot0 = (np.sin(lon0)*np.cos(lat0), -np.cos(lon0)*np.cos(lat0), np.sin(lat0))
        longitude = np.linspace(0, 2*np.pi, 361)
        latitude = np.linspace(-np.pi/2, np.pi/2, 181)
        ptsx = np.outer(np.sin(longitude), np.cos(latitude))
        ptsy = -np.outer(np.cos(longitude), np.cos(latitude))
        ptsz = -np.outer(np.ones_like(longitude), np.sin(latitude))
        cosphi = ptsx*spot0[0]+ptsy*spot0[1]+ptsz*spot0[2]
        cosphi[cosphi > 1] = 1
        cosphi[cosphi < -1] = -1
        sourcemap = np.exp(-np.arccos(cosphi)/sigma0)
        lon, lat = random_deviates_2d(sourcemap, longitude, latitude, npack)
    elif spatialdist.type == 'surface map':
        raise NotImplementedError('surface-map sources need the reference\'s pickled map files; '
                                  'out of scope (SURVEY.md section 2)')
    else:
        assert False, "Can't get here"

    # The reference passes geometry.planet.type == 'Planet' (always True, :126).  With a moon
    # as start point (our extension) the satellite convention of xyz_from_lonlat applies.
    geo = outputs.inputs.geometry
    X_ = xyz_from_lonlat(lon, lat, geo.planet.object == geo.startpoint, spatialdist.exobase)
    outputs.X0['x'] = X_[0, :]
    outputs.X0['y'] = X_[1, :]
    outputs.X0['z'] = X_[2, :]
    outputs.X0['longitude'] = lon
    outputs.X0['latitude'] = lat
    outputs.X0['local_time'] = (lon * 12/np.pi + 12) % 24


def speed_distribution(outputs):
    """Launch speeds in R/s (source_distribution.py:137-189)."""
    speeddist = outputs.inputs.speeddist
    npackets = outputs.npackets
    species = outputs.inputs.options.species

    if speeddist.type.lower() == 'gaussian':
        if speeddist.sigma == 0.:
            v0 = np.zeros(npackets) + speeddist.vprob.value
        else:
            v0 = (outputs.randgen.standard_normal(npackets) * speeddist.sigma.value
                  + speeddist.vprob.value)
    elif speeddist.type == 'sputtering':
        velocity = np.linspace(.1, 50, 5000)
        f_v = sputdist(velocity, speeddist.U.value, speeddist.alpha, speeddist.beta, species)
        v0 = random_deviates_1d(velocity, f_v, npackets)
    elif speeddist.type == 'maxwellian':
        if speeddist.temperature != 0:
            v_th = np.sqrt(2*speeddist.temperature.value*const.K_B/_mass_kg(species)) / 1e3
            velocity = np.linspace(0.1, v_th*5, 5000)
            f_v = MaxwellianDist(velocity, speeddist.temperature.value, species)
            v0 = random_deviates_1d(velocity, f_v, npackets)
        else:
            assert 0, 'Not implemented yet'
    elif speeddist.type == 'flat':
        v0 = (outputs.randgen.random(npackets)*2*speeddist.delv.value
              + speeddist.vprob.value - speeddist.delv.value)
    elif speeddist.type == 'user defined':
        raise InputError('speed_distribution', 'user-defined speed files are out of scope')
    else:
        assert 0, 'Distribtuion does not exist'

    v0 = v0 / outputs.unit_km           # km/s -> R/s
    outputs.X0['v'] = v0
    assert np.all(np.isfinite(v0)), 'Infinite values for v0'
    return v0


def angular_distribution(outputs):
    """Launch directions -> vx, vy, vz (source_distribution.py:192-283)."""
    npackets = outputs.npackets
    angulardist = outputs.inputs.angulardist

    if angulardist.type == 'none':
        return
    elif angulardist.type == 'radial':
        alt = np.zeros(npackets) + np.pi/2.
        az = np.zeros(npackets)
    elif angulardist.type == 'isotropic':
        alt0 = angulardist.altitude
        aa = (np.sin(alt0[0]), np.sin(alt0[1]))
        sinalt = outputs.randgen.random(npackets) * (aa[1] - aa[0]) + aa[0]
        alt = np.arcsin(sinalt)
        az0, az1 = (float(v) for v in angulardist.azimuth)
        m = (az0, az1) if az0 <= az1 else (az1, az0+2*np.pi)
        az = m[0] + (m[1]-m[0])*outputs.randgen.random(npackets)
    elif angulardist.type == '2d':
        alt0 = angulardist.altitude
        aa = (np.cos(alt0[0]), np.cos(alt0[1]))
        cosalt = outputs.randgen.random(npackets) * (aa[1] - aa[0]) + aa[0]
        alt = np.arccos(cosalt)
    else:
        assert 0, 'Angular Distribution not defined.'

    x0, y0, z0 = (outputs.X0[c].values for c in ('x', 'y', 'z'))
    speed = outputs.X0.v.values
    if angulardist.type != '2d':
        v_rad = np.sin(alt)
        v_tan0 = np.cos(alt) * np.cos(az)
        v_tan1 = np.cos(alt) * np.sin(az)
        rad = np.array([x0, y0, z0]).transpose()
        east = np.array([y0, -x0, np.zeros_like(z0)]).transpose()
        north = np.array([-z0*x0, -z0*y0, x0**2+y0**2]).transpose()
        rad = rad/np.linalg.norm(rad, axis=1)[:, np.newaxis]
        east = east/np.linalg.norm(east, axis=1)[:, np.newaxis]
        north = north/np.linalg.norm(north, axis=1)[:, np.newaxis]
        v0 = (v_tan0[:, np.newaxis]*north + v_tan1[:, np.newaxis]*east
              + v_rad[:, np.newaxis]*rad)
        outputs.X0['vx'] = v0[:, 0] * speed
        outputs.X0['vy'] = v0[:, 1] * speed
        outputs.X0['vz'] = v0[:, 2] * speed
        outputs.X0['altitude'] = alt
        outputs.X0['azimuth'] = az
    else:
        v_rad = np.sin(alt)
        v_tan = np.cos(alt)
        rad = np.array([x0, y0]).transpose()
        tan = np.array([y0, -x0]).transpose()
        rad = rad/np.linalg.norm(rad, axis=1)[:, np.newaxis]
        tan = tan/np.linalg.norm(tan, axis=1)[:, np.newaxis]
        v0 = v_tan[:, np.newaxis]*tan + v_rad[:, np.newaxis]*rad
        assert np.all(np.isclose(np.sum(v0**2, axis=1), 1))
        outputs.X0['vx'] = v0[:, 0] * speed
        outputs.X0['vy'] = v0[:, 1] * speed
        outputs.X0['vz'] = np.zeros((npackets, ))
        outputs.X0['altitude'] = alt
        outputs.X0['azimuth'] = 0
        outputs.X0['v_radial'] = v_rad * outputs.X0['v']
        outputs.X0['v_east'] = np.sqrt(outputs.X0['v']**2 - outputs.X0['v_radial']**2)
        outputs.X0['v_north'] = 0
