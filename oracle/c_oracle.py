"""ctypes binding of the C oracle (oracle/c/oracle.c -> oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by nexoclom_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)


class _Forces(C.Structure):
    _fields_ = [('GM', C.c_double), ('vrplanet', C.c_double), ('photo', C.c_double),
                ('lifetime', C.c_double), ('gravity', C.c_int32), ('radpres', C.c_int32),
                ('has_photo', C.c_int32), ('pad_', C.c_int32), ('n_tab', C.c_int64),
                ('v_tab', _dp), ('a_tab', _dp)]


class _Image(C.Structure):
    _fields_ = [('M', C.c_double*9), ('vrplanet', C.c_double), ('apix_cm2', C.c_double),
                ('quantity', C.c_int32), ('n_lines', C.c_int32), ('downcast_f32', C.c_int32),
                ('pad_', C.c_int32), ('nx', C.c_int64), ('nz', C.c_int64),
                ('xedges', _dp), ('zedges', _dp), ('line_n', C.c_int64*4),
                ('line_v', _dp*4), ('line_g', _dp*4)]


class _Bodies(C.Structure):
    _fields_ = [('n_moons', C.c_int32), ('chx_on', C.c_int32), ('gm', C.c_double*4),
                ('radius', C.c_double*4), ('a', C.c_double*4), ('omega', C.c_double*4),
                ('phi', C.c_double*4), ('t0', C.c_double), ('chx_k0', C.c_double),
                ('chx_rho0', C.c_double), ('chx_width', C.c_double), ('chx_height', C.c_double),
                ('chx_omega', C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, '_build', 'liboracle.so')
    if force or not all(os.path.exists(os.path.join(_HERE, '_build', n)) for n in
                        ('liboracle.so', 'liboracle_libm.so', 'liboracle_2r.so')):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return so


def _ptr(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class COracle:
    def __init__(self, libm=False, two_roundings=False):
        """two_roundings: the build whose tableau terms round twice, like NumPy's (the default
        build fuses them, like the kernels)."""
        build()
        name = 'liboracle_libm.so' if libm else 'liboracle_2r.so' if two_roundings else 'liboracle.so'
        self.lib = C.CDLL(os.path.join(_HERE, '_build', name))
        self.lib.ora_integrate_const.restype = C.c_int64
        self.lib.ora_integrate_const_bodies.restype = C.c_int64
        self.lib.ora_integrate_var.restype = C.c_int64
        self.lib.ora_max_threads.restype = C.c_int
        self._keep = []

    def max_threads(self):
        return int(self.lib.ora_max_threads())

    def forces(self, f):
        """f: oracle.np_oracle.Forces (or anything with the same attributes)."""
        v, a = _f64(f.v_tab), _f64(f.a_tab)
        self._keep.append((v, a))
        return _Forces(f.GM, f.vrplanet, 0.0 if f.photo is None else f.photo, f.lifetime,
                       int(f.gravity), int(f.radpres), int(f.photo is not None), 0,
                       len(v), _ptr(v), _ptr(a))

    def image_desc(self, M, vrplanet, apix_cm2, quantity, g_tables, xedges, zedges,
                   downcast=False):
        d = _Image()
        d.M = (C.c_double*9)(*np.asarray(M, float).reshape(9))
        d.vrplanet, d.apix_cm2 = vrplanet, apix_cm2
        d.quantity = 0 if quantity in ('column', 'density') else 1
        d.n_lines = len(g_tables) if d.quantity == 1 else 0
        d.downcast_f32 = int(downcast)
        xe, ze = _f64(xedges), _f64(zedges)
        d.nx, d.nz = len(xe)-1, len(ze)-1
        d.xedges, d.zedges = _ptr(xe), _ptr(ze)
        keep = [xe, ze]
        for k, (v, g) in enumerate(g_tables[:4]):
            v, g = _f64(v), _f64(g)
            keep += [v, g]
            d.line_n[k] = len(v)
            d.line_v[k], d.line_g[k] = _ptr(v), _ptr(g)
        self._keep.append(keep)
        return d

    def state(self, f, x, y, z, vy):
        ff = self.forces(f)
        x, y, z, vy = map(_f64, (x, y, z, vy))
        n = len(x)
        out = [np.empty(n) for _ in range(4)]
        self.lib.ora_state(C.byref(ff), C.c_int64(n), _ptr(x), _ptr(y), _ptr(z), _ptr(vy),
                           *[_ptr(o) for o in out])
        return np.stack(out[:3], axis=1), out[3]

    def rk5(self, f, X0, h, want_delta=False):
        """X0 (N,8) row-major like the reference; returns (N,8) result and delta or None."""
        ff = self.forces(f)
        n = X0.shape[0]
        soa = _f64(X0.T)
        hh = _f64(np.broadcast_to(h, (n,)))
        out = np.empty((8, n))
        delta = np.empty((8, n)) if want_delta else None
        self.lib.ora_rk5_step(C.byref(ff), C.c_int64(n), _ptr(soa), _ptr(hh), _ptr(out),
                              _ptr(delta) if want_delta else None)
        return out.T.copy(), (delta.T.copy() if want_delta else None)

    @staticmethod
    def bodies(b):
        """b: oracle.np_oracle.Bodies."""
        d = _Bodies()
        d.n_moons, d.chx_on = len(b.gm), int(b.chx_on)
        for m in range(len(b.gm)):
            d.gm[m], d.radius[m], d.a[m] = b.gm[m], b.radius[m], b.a[m]
            d.omega[m], d.phi[m] = b.omega[m], b.phi[m]
        d.t0 = b.t0
        d.chx_k0, d.chx_rho0, d.chx_width = b.chx_k0, b.chx_rho0, b.chx_width
        d.chx_height, d.chx_omega = b.chx_height, b.chx_omega
        return d

    def integrate_const(self, f, X0, step, n_iter, outeredge, nrec=0, img=None, threads=1,
                        want_final=True, bodies=None):
        """Returns dict(work, traj (8,nrec,n)|None, final (N,8), steps (N,), image, counts)."""
        ff = self.forces(f)
        n = X0.shape[0]
        soa = _f64(X0.T)
        traj = np.zeros((8, nrec, n)) if nrec else None
        final = np.empty((8, n)) if want_final else None
        steps = np.empty(n, dtype=np.int64)
        image = counts = None
        if img is not None:
            image = np.zeros((img.nx, img.nz))
            counts = np.zeros((img.nx, img.nz), dtype=np.uint64)
        fn, lead = self.lib.ora_integrate_const, (C.byref(ff),)
        if bodies is not None:
            bb = self.bodies(bodies)
            fn, lead = self.lib.ora_integrate_const_bodies, (C.byref(ff), C.byref(bb))
        work = fn(
            *lead, C.c_int64(n), _ptr(soa), C.c_double(step), C.c_int64(n_iter),
            C.c_double(outeredge), _ptr(traj) if nrec else None, C.c_int64(nrec),
            _ptr(final) if want_final else None, steps.ctypes.data_as(C.c_void_p),
            C.byref(img) if img is not None else None,
            _ptr(image) if img is not None else None,
            counts.ctypes.data_as(C.c_void_p) if img is not None else None, C.c_int(threads))
        return dict(work=int(work), traj=traj, final=None if final is None else final.T.copy(),
                    steps=steps, image=image, counts=counts)

    def integrate_var(self, f, X0, resolution, outeredge, max_steps=10**7):
        ff = self.forces(f)
        n = X0.shape[0]
        soa = _f64(X0.T)
        out = np.empty((8, n))
        hs = np.empty(n)
        bad = C.c_int64(0)
        work = self.lib.ora_integrate_var(C.byref(ff), C.c_int64(n), _ptr(soa),
                                          C.c_double(resolution), C.c_double(outeredge),
                                          C.c_int64(max_steps), _ptr(out), _ptr(hs),
                                          C.byref(bad))
        return out.T.copy(), hs, int(work), int(bad.value)

    def image(self, img, x, y, z, vy, frac):
        x, y, z, vy, frac = map(_f64, (x, y, z, vy, frac))
        image = np.zeros((img.nx, img.nz))
        counts = np.zeros((img.nx, img.nz), dtype=np.uint64)
        self.lib.ora_image(C.byref(img), C.c_int64(len(x)), _ptr(x), _ptr(y), _ptr(z), _ptr(vy),
                           _ptr(frac), _ptr(image), counts.ctypes.data_as(C.c_void_p))
        return image, counts

    def math(self, which, x):
        x = _f64(x)
        out = np.empty_like(x)
        self.lib.ora_math_batch(C.c_int({'exp': 0, 'log': 1, 'cube': 2}[which]),
                                C.c_int64(len(x)), _ptr(x), _ptr(out))
        return out
