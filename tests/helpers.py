"""Shared set-up for the parity tests: forces for the BASELINE configurations and seeded X0."""
import numpy as np

from nexoclom_amd.atomicdata import PhotoRate, RadPresConst, gValue
from nexoclom_amd.solarsystem import SSObject, planet_dist
from oracle import np_oracle as O


def mercury_forces(species='Na', taa=1.3, gravity=True, radpres=True, lifetime=0.0):
    """oracle Forces for <species> at Mercury (Output.__init__ set-up, Output.py:105-128)."""
    m = SSObject('Mercury')
    R = m.radius.value                        # km
    r, v = planet_dist(m, taa)
    aplanet = float(r)
    rp = RadPresConst(species, aplanet)
    if lifetime > 0:
        photo = None
    elif lifetime < 0:
        photo = abs(1./lifetime)
    else:
        photo = PhotoRate(species, aplanet).rate.value
    f = O.Forces(GM=m.GM.value/(R*1e3)**3, vrplanet=float(v)/R, gravity=gravity, radpres=radpres,
                 lifetime=lifetime, photo=photo, v_tab=rp.velocity/R, a_tab=rp.accel/R)
    f.aplanet = aplanet
    f.R_km = R
    return f


def set_ctx_forces(ctx, f):
    ctx.set_forces(f.GM, f.vrplanet, f.gravity, f.radpres, f.lifetime, f.photo, f.v_tab, f.a_tab)


def g_tables(species, aplanet, R_km, wavelengths):
    out = []
    for w in wavelengths:
        g = gValue(species, w, aplanet)
        out.append((g.velocity/R_km, g.g))
    return out


def sample_x0(n, seed, endtime, vprob=2.5, delv=2.0, R_km=2440.53, exobase=1.0):
    """Uniform surface / flat speed / isotropic direction, in the reference's draw order
    (source_distribution.py:47-62,169-171,202-212,226-252)."""
    rng = np.random.default_rng(seed)
    sinlat = -1 + 2*rng.random(n)
    lat = np.arcsin(sinlat)
    lon = (0 + 2*np.pi*rng.random(n)) % (2*np.pi)
    x0 = exobase*np.sin(lon)*np.cos(lat)
    y0 = -exobase*np.cos(lon)*np.cos(lat)
    z0 = exobase*np.sin(lat)
    v = (rng.random(n)*2*delv + vprob - delv)/R_km
    alt = np.arcsin(rng.random(n))
    az = 2*np.pi*rng.random(n)
    v_rad, v_t0, v_t1 = np.sin(alt), np.cos(alt)*np.cos(az), np.cos(alt)*np.sin(az)
    rad = np.stack([x0, y0, z0], 1)
    east = np.stack([y0, -x0, np.zeros(n)], 1)
    north = np.stack([-z0*x0, -z0*y0, x0**2+y0**2], 1)
    rad /= np.linalg.norm(rad, axis=1)[:, None]
    east /= np.linalg.norm(east, axis=1)[:, None]
    north /= np.linalg.norm(north, axis=1)[:, None]
    vdir = v_t0[:, None]*north + v_t1[:, None]*east + v_rad[:, None]*rad
    X = np.zeros((n, 8))
    X[:, 0] = endtime
    X[:, 1:4] = rad*exobase
    X[:, 1], X[:, 2], X[:, 3] = x0, y0, z0
    X[:, 4:7] = vdir*v[:, None]
    X[:, 7] = 1.0
    return X


def random_cloud(n, seed, R_km=2440.53, tmax=50000.):
    """Packets scattered through the cloud volume (inside/outside the shadow, near rho = 1, on
    y = 0, with radial velocities inside and outside the g-value table)."""
    rng = np.random.default_rng(seed)
    X = np.zeros((n, 8))
    X[:, 0] = rng.uniform(100, tmax, n)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    X[:, 1:4] = d*rng.uniform(1.0, 6.0, n)[:, None]
    X[:, 4:7] = rng.normal(size=(n, 3))*2.0/R_km
    X[:, 7] = rng.uniform(1e-6, 1.0, n)
    k = n//8
    X[:k, 1], X[:k, 3] = 0.6*np.cos(np.arange(k)), 0.6*np.sin(np.arange(k))    # rho < 1
    X[:k, 2] = np.where(np.arange(k) % 2 == 0, 2.0, -2.0)                      # behind / in front
    X[k:k+4, 1], X[k:k+4, 3], X[k:k+4, 2] = 1.0, 0.0, [1.5, -1.5, 0.0, 3.0]   # rho == 1 exactly
    X[k+4:k+8, 2] = 0.0                                                        # y == 0
    X[k+8:k+12, 5] = np.array([60., -70., 49.5196, -50.6857])/R_km - 9.73/R_km  # table ends
    return X


def image_setup(f, quantity='radiance', dims=(64, 64), center=(0., 0.), width=(8., 8.),
                sublon=0.0, sublat=np.pi/2, species='Na', wavelengths=(5891, 5897)):
    """Everything create_image needs (ModelImage.__init__, ModelImage.py:53-78)."""
    M = O.image_rotation(sublon, sublat)
    xr = (center[0]-width[0]/2, center[0]+width[0]/2)
    zr = (center[1]-width[1]/2, center[1]+width[1]/2)
    xedges = np.linspace(xr[0], xr[1], dims[0]+1)
    zedges = np.linspace(zr[0], zr[1], dims[1]+1)
    R_cm = f.R_km*1e5
    apix = (width[0]/dims[0])*(width[1]/dims[1])*R_cm**2
    gt = g_tables(species, f.aplanet, f.R_km, wavelengths) if quantity == 'radiance' else []
    return dict(M=M, xrange=xr, zrange=zr, xedges=xedges, zedges=zedges, apix=apix,
                g_tables=gt, quantity=quantity, dims=list(dims))
