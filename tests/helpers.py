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


def collinear_point(GM, gm, a, omega):
    """L1 of THIS model (planet held fixed at the origin, moon on a prescribed circle: there is
    no indirect term, so no triangular points either) and the growth rate of its unstable mode.
    Rotating frame, x along planet -> moon: effective potential
    U = GM/r + gm/|r - r_m| - omega^2 r^2 / 2  (GM, gm negative as in the C ABI).
    Equilibrium: |GM|/x^2 - |gm|/(a - x)^2 = omega^2 x.  Linearised: d2(xi)/dt2 - 2 omega
    d(eta)/dt = A xi, d2(eta)/dt2 + 2 omega d(xi)/dt = B eta with A = omega^2 + 2 K, B = omega^2 -
    K, K = |GM|/x^3 + |gm|/(a - x)^3;  lambda^4 + (4 omega^2 - A - B) lambda^2 + A B = 0."""
    M, m = -GM, -gm
    lo, hi = 0.5*a, a*(1 - 1e-9)
    f = lambda x: M/x**2 - m/(a - x)**2 - omega**2*x          # noqa: E731
    for _ in range(200):
        mid = 0.5*(lo + hi)
        lo, hi = (mid, hi) if f(mid) > 0 else (lo, mid)
    x = 0.5*(lo + hi)
    K = M/x**3 + m/(a - x)**3
    A, B = omega**2 + 2*K, omega**2 - K
    b = 4*omega**2 - A - B
    lam2 = (-b + np.sqrt(b*b - 4*A*B))/2
    lam = np.sqrt(lam2)
    # unstable eigenvector (xi, eta): (lam^2 - A) xi = 2 omega lam eta
    return x, lam, (lam2 - A)/(2*omega*lam)



def collinear_case():
    """Five packets around the inner collinear point of a planet + one moon of mass ratio 0.01:
    one exactly on it with the co-rotating velocity, four displaced along the unstable
    eigenvector of the linearised rotating-frame equations (two opposite, one doubled, one ten
    times as far)."""
    GM, mu, a = -1.0e-6, 0.01, 6.0
    gm = mu*GM
    omega = np.sqrt(-GM/a**3)
    x1, lam, slope = collinear_point(GM, gm, a, omega)
    assert 0.8*a*(1 - (mu/3)**(1/3)) < x1 < a and 2.0 < lam/omega < 3.5
    period = 2*np.pi/omega
    step, n_iter = period/4000, 1334                   # a third of an orbit: lambda t = 6.1
    T = step*n_iter
    phi = 0.3                                          # moon's phase at t_remaining = 0
    theta0 = phi - omega*T                             # ... and at the start
    ex = np.array([-np.sin(theta0), np.cos(theta0), 0.])        # planet -> moon
    ey = np.array([-np.cos(theta0), -np.sin(theta0), 0.])       # 90 degrees ahead
    eps = np.array([0., 1e-7, -1e-7, 2e-7, 1e-6])*a
    X0 = np.zeros((len(eps), 8))
    for i, d in enumerate(eps):
        xi, eta = d, d*slope
        pos = (x1 + xi)*ex + eta*ey
        # rotating-frame velocity of the unstable mode (lam xi, lam eta) + co-rotation omega z x r
        vel = lam*(xi*ex + eta*ey) + omega*np.cross([0., 0., 1.], pos)
        X0[i] = [T, *pos, *vel, 1.0]
    return dict(GM=GM, gm=gm, a=a, omega=omega, phi=phi, T=T, step=step, n_iter=n_iter, X0=X0,
                eps=eps, x1=x1, lam=lam, slope=slope, theta0=theta0, period=period)


def check_collinear_run(case, traj):
    """traj (8, n_iter + 1, 5) of collinear_case(): the packet on the point stays, the displaced
    ones leave as exp(lambda t) along the unstable direction."""
    a, lam, slope, eps = case['a'], case['lam'], case['slope'], case['eps']
    k = np.arange(case['n_iter'] + 1)
    th = case['theta0'] + case['omega']*case['step']*k
    ux, uy = np.stack([-np.sin(th), np.cos(th)]), np.stack([-np.cos(th), -np.sin(th)])
    xi = traj[1]*ux[0][:, None] + traj[2]*ux[1][:, None] - case['x1']
    eta = traj[1]*uy[0][:, None] + traj[2]*uy[1][:, None]
    assert np.all(traj[7] == 1.0) and np.all(traj[3] == 0.0)
    # the packet at the point itself: what is left is the seed of the instability from rounding
    # and the integrator's error, far below the displaced packets' excursions
    assert np.abs(xi[:, 0]).max() < 1e-11*a and np.abs(eta[:, 0]).max() < 1e-11*a
    t = case['step']*k
    for i in range(1, len(eps)):
        grow = (xi[:, i] - xi[:, 0])/eps[i]
        late = t > 0.15*case['period']                 # the decaying companion modes have died
        rate = np.polyfit(t[late], np.log(grow[late]), 1)[0]
        assert abs(rate/lam - 1) < 1e-3, (i, rate/lam)
        np.testing.assert_allclose(grow[-1], np.exp(lam*case['T']), rtol=5e-3)
        np.testing.assert_allclose((eta[late, i] - eta[late, 0])/(xi[late, i] - xi[late, 0]),
                                   slope, rtol=2e-3)
    # opposite displacements run away in opposite directions, twice the displacement twice as far
    np.testing.assert_allclose(xi[-1, 2] - xi[-1, 0], -(xi[-1, 1] - xi[-1, 0]), rtol=1e-3)
    np.testing.assert_allclose(xi[-1, 3] - xi[-1, 0], 2*(xi[-1, 1] - xi[-1, 0]), rtol=1e-3)
