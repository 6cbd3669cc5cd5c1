"""ctypes binding of libnexoclom_hip.so (include/nexoclom_hip.h) -- the only door to the GPU.

There is no CPU fallback: if the library is missing or no gfx950 device is visible, every
entry point raises.  Arrays cross the boundary as C-contiguous float64 NumPy buffers.
"""
import ctypes as C
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NEXOCLOM_HIP_LIB: load another build of the same library (installation elsewhere, experiments)
LIB_PATH = os.environ.get('NEXOCLOM_HIP_LIB') or os.path.join(_HERE, 'lib', 'libnexoclom_hip.so')
_dp = C.POINTER(C.c_double)
NXC_MAX_LINES = 4
NXC_RUN_IMAGE = 1
NXC_UNIQUE_ID_BYTES = 128

# every symbol include/nexoclom_hip.h declares
EXPORTS = ('nxc_abi_version', 'nxc_device_count', 'nxc_last_error_string', 'nxc_create',
           'nxc_destroy', 'nxc_device_name', 'nxc_synchronize', 'nxc_set_forces',
           'nxc_set_image', 'nxc_state', 'nxc_rk5_step', 'nxc_packets_upload', 'nxc_image_clear',
           'nxc_image_download', 'nxc_counters_get', 'nxc_last_kernel_ms', 'nxc_integrate_const',
           'nxc_integrate_const_async', 'nxc_integrate_var', 'nxc_image_accumulate',
           'nxc_image_accumulate_f32',
           'nxc_comm_unique_id', 'nxc_comm_init', 'nxc_comm_destroy', 'nxc_image_allreduce',
           'nxc_allreduce_max_f64', 'nxc_barrier', 'nxc_math_batch', 'nxc_los_accumulate',
           'nxc_los_accumulate_f32', 'nxc_packets_sample',
           'nxc_set_bounce', 'nxc_set_first_index', 'nxc_set_bodies',
           'nxc_integrate_const_rows', 'nxc_rows_fetch', 'nxc_rows_fetch_f32', 'nxc_device_bus_id',
           'nxc_allreduce_sum_f64', 'nxc_rows_build', 'nxc_rows_info', 'nxc_rows_download',
           'nxc_rows_free', 'nxc_image_accumulate_rows', 'nxc_los_accumulate_rows', 'nxc_mem_info',
           'nxc_stream_copy_gbs', 'nxc_shader_clock_mhz', 'nxc_pcg64_uniforms',
           'nxc_integrate_const_streamed', 'nxc_image_mode', 'nxc_allreduce_f64',
           'nxc_comm_set_timeout', 'nxc_comm_abort', 'nxc_comm_request_abort',
           'nxc_comm_test_stall', 'nxc_packets_upload_pieces')
ABI_VERSION = 3


NXC_ERR_HIP, NXC_ERR_ARG, NXC_ERR_NO_DEVICE, NXC_ERR_RCCL, NXC_ERR_STATE, NXC_ERR_NOMEM, \
    NXC_ERR_INCOMPLETE = -1, -2, -3, -4, -5, -6, -7


class HipError(RuntimeError):
    """An nxc_* call failed; ``code`` is its NXC_ERR_* status (None when raised by the binding)."""

    def __init__(self, message, code=None):
        super().__init__(message)
        self.code = code


class nxc_forces(C.Structure):
    _fields_ = [('GM', C.c_double), ('vrplanet', C.c_double), ('photo', C.c_double),
                ('lifetime', C.c_double), ('gravity', C.c_int32), ('radpres', C.c_int32),
                ('has_photo', C.c_int32), ('reserved', C.c_int32), ('n_tab', C.c_int64),
                ('v_tab', _dp), ('a_tab', _dp)]


class nxc_image_desc(C.Structure):
    _fields_ = [('M', C.c_double*9), ('vrplanet', C.c_double), ('apix_cm2', C.c_double),
                ('quantity', C.c_int32), ('n_lines', C.c_int32), ('downcast_f32', C.c_int32),
                ('reserved', C.c_int32), ('nx', C.c_int64), ('nz', C.c_int64),
                ('xedges', _dp), ('zedges', _dp), ('line_n', C.c_int64*NXC_MAX_LINES),
                ('line_v', _dp*NXC_MAX_LINES), ('line_g', _dp*NXC_MAX_LINES)]


class nxc_los_desc(C.Structure):
    _fields_ = [('dphi', C.c_double), ('sin_dphi', C.c_double), ('sin_2dphi', C.c_double),
                ('cos_threshold', C.c_double), ('vrplanet', C.c_double), ('unit_cm', C.c_double),
                ('n_lines', C.c_int32), ('reserved', C.c_int32),
                ('line_n', C.c_int64*NXC_MAX_LINES), ('line_v', _dp*NXC_MAX_LINES),
                ('line_g', _dp*NXC_MAX_LINES), ('n_ladder', C.c_int64), ('ladder', _dp)]


class nxc_source_desc(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ('endtime', 'exobase', 'sinlat0', 'sinlat1', 'lon0', 'lon1', 'vprob', 'vwidth',
                 'unit_km', 'sinalt0', 'sinalt1', 'az0', 'az1')] + \
               [(k, C.c_int32) for k in ('random_time', 'speed_type', 'angular_type', 'is_planet')] + \
               [('seed', C.c_uint64), ('first_index', C.c_int64), ('spatial_type', C.c_int32),
                ('reserved', C.c_int32), ('n_speed', C.c_int64), ('speed_cdf', _dp),
                ('speed_v', _dp), ('map_nlon', C.c_int64), ('map_nlat', C.c_int64), ('map', _dp),
                ('generator', C.c_int32), ('reserved2', C.c_int32), ('pcg_state', C.c_uint64*2),
                ('pcg_inc', C.c_uint64*2), ('pcg_n', C.c_int64), ('pcg_row0', C.c_int64),
                ('dest_offset', C.c_int64), ('dest_total', C.c_int64)]


class nxc_bounce_desc(C.Structure):
    _fields_ = [('GM', C.c_double), ('unit_km', C.c_double), ('accomfactor', C.c_double),
                ('stickcoef', C.c_double), ('A', C.c_double*3), ('t0', C.c_double),
                ('t1', C.c_double), ('temp_dependent', C.c_int32), ('reserved', C.c_int32),
                ('nx', C.c_int64), ('ny', C.c_int64), ('tx', _dp), ('ty', _dp), ('coef', _dp),
                ('seed', C.c_uint64)]


class nxc_bodies_desc(C.Structure):
    _fields_ = [('n_moons', C.c_int32), ('chx_on', C.c_int32), ('gm', C.c_double*4),
                ('radius', C.c_double*4), ('a', C.c_double*4), ('omega', C.c_double*4),
                ('phi', C.c_double*4), ('t0', C.c_double), ('chx_k0', C.c_double),
                ('chx_rho0', C.c_double), ('chx_width', C.c_double), ('chx_height', C.c_double),
                ('chx_omega', C.c_double)]


class nxc_counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                ('particle_steps', 'samples', 'samples_binned', 'nonfinite', 'bad_step',
                 'neg_frac', 'unfinished', 'wave_trips')]


_lib = None


def load_library():
    """dlopen the in-tree library (building nothing); raises HipError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipError(f'{LIB_PATH} not found: build it with `python -m nexoclom_amd.build` '
                       '(there is no CPU fallback for the hot path)')
    # several processes on one node (RCCL, shared device memory): this pool's host driver only
    # supports dmabuf IPC; without the setting ncclCommInitRank fails in hipIpcGetMemHandle.  Read
    # by the runtime when it initialises, i.e. at the first HIP call -- none has been made yet.
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    lib = C.CDLL(LIB_PATH)
    lib.nxc_last_error_string.restype = C.c_char_p
    missing = [name for name in EXPORTS if not hasattr(lib, name)]
    # (an explicitly named build -- experiments comparing library versions -- may lack the newest
    # entry points; the in-tree library may not)
    if missing and not os.environ.get('NEXOCLOM_HIP_LIB'):
        raise HipError(f'{LIB_PATH} lacks {missing}: rebuild with `python -m nexoclom_amd.build`')
    if lib.nxc_abi_version() != ABI_VERSION:
        raise HipError(f'libnexoclom_hip.so has ABI version {lib.nxc_abi_version()}, this binding '
                       f'is written for {ABI_VERSION}: rebuild with `python -m nexoclom_amd.build`')
    _lib = lib
    return lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def pcg64_words(seed):
    """(state, inc) of numpy.random.default_rng(seed)'s bit generator as {high, low} 64-bit words
    (what nxc_source_desc.pcg_state / pcg_inc take)."""
    st = np.random.PCG64(seed).state['state']
    mask = (1 << 64) - 1
    return ((C.c_uint64*2)(st['state'] >> 64, st['state'] & mask),
            (C.c_uint64*2)(st['inc'] >> 64, st['inc'] & mask))


def _p(a):
    return a.ctypes.data_as(_dp)


def device_count():
    lib = load_library()
    n = C.c_int(0)
    rc = lib.nxc_device_count(C.byref(n))
    if rc != 0:
        return 0
    return n.value


class RowStore:
    """Compact trajectory rows that stay in HBM (an ``nxc_rows``): the nine columns time, x, y, z,
    vx, vy, vz, frac, lossfrac and the packet-index column of ``total`` live records, float32 /
    int32 when ``narrow`` (what the reference's save() stores) or float64 / int64.  Views of it
    (row ranges) feed the image and line-of-sight kernels without touching the host."""

    def __init__(self, ctx, handle):
        self.ctx, self._r = ctx, handle
        self.owners = weakref.WeakSet()       # the Outputs whose rows these are
        self._lock = threading.RLock()        # a file writer may be reading while the store is spilled
        total, f32 = C.c_int64(0), C.c_int32(0)
        ctx._check(ctx.lib.nxc_rows_info(handle, C.byref(total), C.byref(f32)))
        self.total, self.narrow = int(total.value), bool(f32.value)

    @property
    def nbytes(self):
        return self.total * (9*4 + 4 if self.narrow else 9*8 + 8)

    def download(self, first=0, count=None, index=True):
        """(rows (9, count), index (count,) | None) of the row range as host arrays."""
        count = self.total - first if count is None else int(count)
        rows = np.empty((9, count), dtype=np.float32 if self.narrow else np.float64)
        idx = np.empty(count, dtype=np.int32 if self.narrow else np.int64) if index else None
        with self._lock:
            if self._r is None:
                raise HipError('the row store has been freed')
            self.ctx._check(self.ctx.lib.nxc_rows_download(
                self.ctx._h, self._r, C.c_int64(first), C.c_int64(count),
                rows.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p) if index else None))
        return rows, idx

    def free(self):
        with self._lock:
            if self._r is not None and getattr(self.ctx, '_h', None):
                self.ctx.lib.nxc_rows_free(self.ctx._h, self._r)
            self._r = None

    def spill(self):
        """Leave HBM: every owner first takes its rows to the host."""
        for owner in list(self.owners):
            owner._spill()
        self.free()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One device + stream + tables + resident packets/image (an ``nxc_handle``)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self._h = C.c_void_p()
        self._check(self.lib.nxc_create(C.c_int(device), C.byref(self._h)))
        self.device = device
        self.n_packets = 0
        self.image_shape = None
        self._stores = []              # resident RowStores, oldest first (weak references)

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            msg = self.lib.nxc_last_error_string()
            raise HipError(f'nexoclom_hip error {rc}: {msg.decode() if msg else "?"}', rc)

    def make_room(self, need):
        """Before ``need`` bytes of row store (+ scratch) are allocated: spill the oldest resident
        stores to their owners' host memory until they fit (their Outputs keep working from the
        host copy)."""
        alive = [ref for ref in self._stores if ref() is not None and ref()._r is not None]
        self._stores = alive
        while alive and self.mem_info()[0] < need + (1 << 30):
            alive.pop(0)().spill()

    def close(self):
        for ref in getattr(self, '_stores', ()):
            store = ref()
            if store is not None:
                store.free()
        if getattr(self, '_h', None) is not None and self._h:
            self.lib.nxc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._check(self.lib.nxc_device_name(self._h, buf, C.c_int(256)))
        return buf.value.decode()

    def bus_id(self):
        buf = C.create_string_buffer(64)
        self._check(self.lib.nxc_device_bus_id(self._h, buf, C.c_int(64)))
        return buf.value.decode()

    def synchronize(self):
        try:
            self._check(self.lib.nxc_synchronize(self._h))
        except HipError as exc:
            if exc.code == NXC_ERR_INCOMPLETE:
                self.n_packets = 0
            raise

    def mem_info(self):
        """(free, total) bytes of device memory."""
        free, total = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.nxc_mem_info(self._h, C.byref(free), C.byref(total)))
        return int(free.value), int(total.value)

    # -- set-up -----------------------------------------------------------------------------
    def set_forces(self, GM, vrplanet, gravity=True, radpres=True, lifetime=0.0, photo=None,
                   v_tab=None, a_tab=None):
        f = nxc_forces()
        f.GM, f.vrplanet = float(GM), float(vrplanet)
        f.photo = 0.0 if photo is None else float(photo)
        f.lifetime = float(lifetime)
        f.gravity, f.radpres, f.has_photo = int(bool(gravity)), int(bool(radpres)), int(photo is not None)
        if radpres:
            v, a = _f64(v_tab), _f64(a_tab)
            if v.shape != a.shape or v.ndim != 1:
                raise ValueError('v_tab and a_tab must be 1-D arrays of equal length')
            f.n_tab, f.v_tab, f.a_tab = len(v), _p(v), _p(a)
        self._check(self.lib.nxc_set_forces(self._h, C.byref(f)))

    def set_image(self, M, vrplanet, apix_cm2, quantity, xedges, zedges, g_tables=(),
                  downcast_f32=False):
        d = nxc_image_desc()
        d.M = (C.c_double*9)(*np.asarray(M, dtype=float).reshape(9))
        d.vrplanet, d.apix_cm2 = float(vrplanet), float(apix_cm2)
        if quantity in ('column', 'density'):
            d.quantity = 0
        elif quantity in ('radiance', 'difrad'):
            d.quantity = 1
        else:
            raise ValueError(f'{quantity} is invalid.')
        d.downcast_f32 = int(bool(downcast_f32))
        xe, ze = _f64(xedges), _f64(zedges)
        d.nx, d.nz = len(xe)-1, len(ze)-1
        d.xedges, d.zedges = _p(xe), _p(ze)
        keep = [xe, ze]
        if d.quantity == 1:
            if len(g_tables) > NXC_MAX_LINES:
                raise ValueError('too many emission lines')
            d.n_lines = len(g_tables)
            for k, (v, g) in enumerate(g_tables):
                v, g = _f64(v), _f64(g)
                keep += [v, g]
                d.line_n[k], d.line_v[k], d.line_g[k] = len(v), _p(v), _p(g)
        self._check(self.lib.nxc_set_image(self._h, C.byref(d)))
        self.image_shape = (int(d.nx), int(d.nz))

    def set_bounce(self, cfg):
        """cfg: dict from nexoclom_amd.surface.bounce_config, or None for perfect sticking."""
        if cfg is None:
            self._check(self.lib.nxc_set_bounce(self._h, None))
            return
        d = nxc_bounce_desc()
        d.GM, d.unit_km = cfg['GM'], cfg['unit_km']
        d.accomfactor, d.stickcoef = cfg['accomfactor'], cfg['stickcoef']
        d.A = (C.c_double*3)(*cfg['A'])
        d.t0, d.t1 = cfg['t0'], cfg['t1']
        d.temp_dependent = int(cfg['temp_dependent'])
        tx, ty, coef = _f64(cfg['tx']), _f64(cfg['ty']), _f64(cfg['coef'])
        d.nx, d.ny = len(tx), len(ty)
        d.tx, d.ty, d.coef = _p(tx), _p(ty), _p(coef)
        d.seed = int(cfg['seed']) & 0xffffffffffffffff
        self._check(self.lib.nxc_set_bounce(self._h, C.byref(d)))

    def set_bodies(self, cfg):
        """cfg: dict(moons=[dict(gm, radius, a, omega, phi), ...], t0, chx=dict(k0, rho0, width,
        height, omega)|None) in model units (R, s), or None to clear.  Extension beyond the
        reference (include/nexoclom_hip.h, nxc_bodies_desc)."""
        if cfg is None:
            self._check(self.lib.nxc_set_bodies(self._h, None))
            return
        d = nxc_bodies_desc()
        moons = cfg.get('moons', [])
        if len(moons) > 4:
            raise HipError('at most 4 moons')
        d.n_moons = len(moons)
        for m, mo in enumerate(moons):
            d.gm[m], d.radius[m], d.a[m] = mo['gm'], mo['radius'], mo['a']
            d.omega[m], d.phi[m] = mo['omega'], mo['phi']
        d.t0 = cfg['t0']
        chx = cfg.get('chx')
        d.chx_on = int(chx is not None)
        if chx is not None:
            d.chx_k0, d.chx_rho0, d.chx_width = chx['k0'], chx['rho0'], chx['width']
            d.chx_height, d.chx_omega = chx['height'], chx.get('omega', 0.0)
        self._check(self.lib.nxc_set_bodies(self._h, C.byref(d)))

    def set_first_index(self, first_index):
        self._check(self.lib.nxc_set_first_index(self._h, C.c_int64(int(first_index))))

    # -- a-2 / a-1 --------------------------------------------------------------------------
    def state(self, x, y, z, vy):
        x, y, z, vy = map(_f64, (x, y, z, vy))
        n = len(x)
        out = [np.empty(n) for _ in range(4)]
        self._check(self.lib.nxc_state(self._h, C.c_int64(n), _p(x), _p(y), _p(z), _p(vy),
                                       *[_p(o) for o in out]))
        return np.stack(out[:3], axis=1), out[3]

    def rk5_step(self, X0, h, want_delta=False):
        """X0 (N,8) as the reference's rk5 takes it; returns ((N,8) result, (N,8) delta|None)."""
        X0 = np.asarray(X0, dtype=np.float64)
        n = X0.shape[0]
        soa = _f64(X0.T)
        hh = _f64(np.broadcast_to(np.asarray(h, dtype=np.float64), (n,)))
        out = np.empty((8, n))
        delta = np.empty((8, n)) if want_delta else None
        self._check(self.lib.nxc_rk5_step(self._h, C.c_int64(n), _p(soa), _p(hh), _p(out),
                                          _p(delta) if want_delta else None))
        return np.ascontiguousarray(out.T), (np.ascontiguousarray(delta.T) if want_delta else None)

    # -- resident data ----------------------------------------------------------------------
    def upload_packets(self, X0):
        """X0 (N,8) row-major or an (8,N) SoA array flagged by ``soa=True`` via upload_soa."""
        X0 = np.asarray(X0, dtype=np.float64)
        return self.upload_soa(_f64(X0.T))

    def upload_soa(self, soa):
        soa = _f64(soa)
        assert soa.ndim == 2 and soa.shape[0] == 8
        self._check(self.lib.nxc_packets_upload(self._h, C.c_int64(soa.shape[1]), _p(soa)))
        self.n_packets = soa.shape[1]

    def upload_soa_pieces(self, pieces):
        """The resident set = the (8, n_p) arrays of ``pieces`` one after the other, copied
        column by column from where they are (no concatenated host copy)."""
        pieces = [_f64(p) for p in pieces]
        assert all(p.ndim == 2 and p.shape[0] == 8 for p in pieces)
        counts = (C.c_int64*len(pieces))(*[p.shape[1] for p in pieces])
        ptrs = (_dp*len(pieces))(*[_p(p) for p in pieces])
        self._check(self.lib.nxc_packets_upload_pieces(self._h, C.c_int32(len(pieces)), counts, ptrs))
        self.n_packets = int(sum(p.shape[1] for p in pieces))

    def sample_packets(self, n, seed, first_index=0, download=False, speed_table=None,
                       surface_map=None, pcg64=None, piece=None, **src):
        """Draw n initial states on the device (nxc_packets_sample).  ``src``: the scalar fields
        of nxc_source_desc except seed/first_index; ``speed_table`` = (cdf, speeds [km/s]) for
        speed_type 2; ``surface_map`` = density array [nlon, nlat] for spatial_type 1 (see
        Output.source_desc).  ``pcg64 = (npackets, row0)``: the reference's own seeded stream --
        rows row0 .. row0 + n - 1 of the npackets-long vectors default_rng(seed) would draw.
        ``piece = (offset, total)``: the n packets are part of a resident set of ``total`` that
        several calls fill in ascending order."""
        d = nxc_source_desc()
        for k, v in src.items():
            setattr(d, k, v)
        d.seed = int(seed) & 0xffffffffffffffff
        d.first_index = int(first_index)
        if pcg64 is not None:
            d.generator, d.pcg_n, d.pcg_row0 = 1, int(pcg64[0]), int(pcg64[1])
            d.pcg_state, d.pcg_inc = pcg64_words(seed)
        if piece is not None:
            d.dest_offset, d.dest_total = int(piece[0]), int(piece[1])
        keep = []
        if speed_table is not None:
            cdf, speeds = _f64(speed_table[0]), _f64(speed_table[1])
            if cdf.shape != speeds.shape or cdf.ndim != 1:
                raise ValueError('speed_table must be two 1-D arrays of equal length')
            keep += [cdf, speeds]
            d.n_speed, d.speed_cdf, d.speed_v = len(cdf), _p(cdf), _p(speeds)
        if surface_map is not None:
            dens = _f64(surface_map)
            keep.append(dens)
            d.map_nlon, d.map_nlat, d.map = dens.shape[0], dens.shape[1], _p(dens)
        out = np.empty((8, int(n))) if download else None
        self._check(self.lib.nxc_packets_sample(self._h, C.byref(d), C.c_int64(int(n)),
                                                _p(out) if download else None))
        self.n_packets = int(n) if piece is None else int(piece[1])
        return out

    def image_clear(self):
        self._check(self.lib.nxc_image_clear(self._h))

    def image_download(self):
        nx, nz = self.image_shape
        image = np.empty((nx, nz))
        counts = np.empty((nx, nz), dtype=np.uint64)
        self._check(self.lib.nxc_image_download(self._h, _p(image),
                                                counts.ctypes.data_as(C.POINTER(C.c_uint64))))
        return image, counts

    def counters(self):
        c = nxc_counters()
        self._check(self.lib.nxc_counters_get(self._h, C.byref(c)))
        return {k: int(getattr(c, k)) for k, _ in nxc_counters._fields_ if k != 'wave_trips'}

    def wave_trips(self):
        """Trips of a wave through the persistent step loop in the last integrate call: a
        measurement (it depends on how the lanes happened to be refilled), kept out of
        ``counters()`` -- whose entries are results and repeat exactly."""
        c = nxc_counters()
        self._check(self.lib.nxc_counters_get(self._h, C.byref(c)))
        return int(c.wave_trips)

    def last_kernel_ms(self):
        ms = C.c_float(0)
        self._check(self.lib.nxc_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    # -- a-3 / a-4 --------------------------------------------------------------------------
    def integrate_const(self, step, n_iter, outeredge, image=False, nrec=0, want_final=False,
                        want_steps=False):
        """Constant-step driver over the resident packets.

        nrec > 0 returns the trajectory as an (8, nrec, N) array (lock-step kernel); otherwise the
        persistent lane-refill kernel runs.  Returns dict(traj, final (N,8), steps)."""
        n = self.n_packets
        traj = np.empty((8, nrec, n)) if nrec else None
        final = np.empty((8, n)) if want_final else None
        steps = np.empty(n, dtype=np.int64) if want_steps else None
        self._check(self.lib.nxc_integrate_const(
            self._h, C.c_double(step), C.c_int64(n_iter), C.c_double(outeredge),
            C.c_uint32(NXC_RUN_IMAGE if image else 0), _p(traj) if nrec else None,
            C.c_int64(nrec), _p(final) if want_final else None,
            steps.ctypes.data_as(C.POINTER(C.c_int64)) if want_steps else None))
        return dict(traj=traj, final=None if final is None else np.ascontiguousarray(final.T),
                    steps=steps)

    def integrate_const_rows(self, step, n_iter, outeredge, narrow=False, resident=False):
        """Trajectories as Output.save() keeps them: only the records with frac > 0,
        packet-major.  Returns dict(lengths (N,) int64, rows (9, total): the 8 state columns and
        lossfrac); narrow=True delivers them as float32 (save()'s down-cast, done on the device).
        resident=True leaves them in HBM instead: dict(lengths, store: RowStore)."""
        n = self.n_packets
        lengths = np.empty(n, dtype=np.int64)
        total = C.c_int64(0)
        self._check(self.lib.nxc_integrate_const_rows(
            self._h, C.c_double(step), C.c_int64(n_iter), C.c_double(outeredge),
            lengths.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(total)))
        if resident:
            # records on their way (10 slots) + the columns that stay (9 + index)
            self.make_room(int(total.value) * 20 * (4 if narrow else 8))
            handle = C.c_void_p()
            self._check(self.lib.nxc_rows_build(self._h, C.c_int(int(narrow)), C.byref(handle)))
            store = RowStore(self, handle)
            self._stores.append(weakref.ref(store))
            return dict(lengths=lengths, store=store)
        rows = np.empty((9, int(total.value)), dtype=np.float32 if narrow else np.float64)
        if narrow:
            self._check(self.lib.nxc_rows_fetch_f32(
                self._h, rows.ctypes.data_as(C.POINTER(C.c_float)) if total.value else None))
        else:
            self._check(self.lib.nxc_rows_fetch(self._h, _p(rows) if total.value else None))
        return dict(lengths=lengths, rows=rows)

    def integrate_const_async(self, step, n_iter, outeredge, image=True):
        self._check(self.lib.nxc_integrate_const_async(
            self._h, C.c_double(step), C.c_int64(n_iter), C.c_double(outeredge),
            C.c_uint32(NXC_RUN_IMAGE if image else 0)))

    def integrate_const_streamed(self, soa, step, n_iter, outeredge, image=True, pieces=16):
        """Upload the (8, N) host array and integrate it in one pipelined pass (the next piece
        crosses PCIe while the current one is integrated).  Asynchronous: ``synchronize()`` before
        touching ``soa`` or reading results -- it raises HipError (code NXC_ERR_INCOMPLETE) when
        the kernel gave up waiting for its queue, in which case nothing of the pass is valid."""
        soa = _f64(soa)
        assert soa.ndim == 2 and soa.shape[0] == 8
        self._keep = soa                      # the copies read it until the stream is drained
        self._check(self.lib.nxc_integrate_const_streamed(
            self._h, C.c_int64(soa.shape[1]), _p(soa), C.c_int32(pieces), C.c_double(step),
            C.c_int64(n_iter), C.c_double(outeredge), C.c_uint32(NXC_RUN_IMAGE if image else 0)))
        self.n_packets = soa.shape[1]

    def integrate_var(self, resolution, outeredge, max_steps=10**6):
        n = self.n_packets
        final = np.empty((8, n))
        hs = np.empty(n)
        self._check(self.lib.nxc_integrate_var(self._h, C.c_double(resolution),
                                               C.c_double(outeredge), C.c_int64(max_steps),
                                               _p(final), _p(hs)))
        return np.ascontiguousarray(final.T), hs

    # -- a-6..a-8 ---------------------------------------------------------------------------
    def image_accumulate(self, x, y, z, vy, frac):
        """Bin stored samples.  Five float32 columns (an Output as save() keeps it) go to the
        device as they are and are widened there; anything else is taken as float64."""
        cols = (x, y, z, vy, frac)
        if all(getattr(c, 'dtype', None) == np.float32 for c in cols):
            cols = [np.ascontiguousarray(c) for c in cols]
            ptr = [c.ctypes.data_as(C.POINTER(C.c_float)) for c in cols]
            self._check(self.lib.nxc_image_accumulate_f32(self._h, C.c_int64(len(cols[0])), *ptr))
            return
        x, y, z, vy, frac = map(_f64, cols)
        self._check(self.lib.nxc_image_accumulate(self._h, C.c_int64(len(x)), _p(x), _p(y), _p(z),
                                                  _p(vy), _p(frac)))

    IMAGE_MODES = {'auto': 0, 'atomics': 1, 'tiles': 2}

    def image_mode(self, mode='auto', tile_pixels=0, slab_samples=0):
        """How stored samples reach the image: 'atomics' (one global atomic pair per binned
        sample), 'tiles' (filed by image tile, summed in LDS, handed over once per pixel), or
        'auto' (tiles from 2^17 samples on).  Packet counts are identical either way."""
        self._check(self.lib.nxc_image_mode(self._h, C.c_int(self.IMAGE_MODES.get(mode, mode)),
                                            C.c_int(int(tile_pixels)),
                                            C.c_int64(int(slab_samples))))

    def image_accumulate_rows(self, store, first=0, count=None):
        """Bin rows [first, first + count) of a RowStore: no host round trip."""
        count = store.total - first if count is None else int(count)
        if store._r is None:
            raise HipError('the row store has been freed')
        self._check(self.lib.nxc_image_accumulate_rows(self._h, store._r, C.c_int64(first),
                                                       C.c_int64(count)))

    # -- f-1: spacecraft lines of sight ------------------------------------------------------
    def los_accumulate(self, dphi, sin_dphi, sin_2dphi, cos_threshold, vrplanet, unit_cm, g_tables,
                       ladder, sc, x=None, y=None, z=None, vy=None, frac=None, index=None,
                       n_index=0, used_cap=0, rows=None):
        """sc: (8, S) array x,y,z,xbore,ybore,zbore,dist_from_plan,ladder_len.  Samples: five host
        columns (+ index), or ``rows = (RowStore, first, count, index_shift)`` for rows that are
        already in HBM.  Returns dict(radiance, npackets, included|None, used (2, m)|None,
        n_used)."""
        d = nxc_los_desc()
        d.dphi, d.sin_dphi, d.sin_2dphi, d.cos_threshold = dphi, sin_dphi, sin_2dphi, cos_threshold
        d.vrplanet, d.unit_cm = float(vrplanet), float(unit_cm)
        keep = []
        d.n_lines = len(g_tables)
        for k, (v, g) in enumerate(g_tables):
            v, g = _f64(v), _f64(g)
            keep += [v, g]
            d.line_n[k], d.line_v[k], d.line_g[k] = len(v), _p(v), _p(g)
        lad = _f64(ladder)
        d.n_ladder, d.ladder = len(lad), _p(lad)
        sc = _f64(sc)
        S = sc.shape[1]
        if rows is not None:
            store, first, count, shift = rows
            if store._r is None:
                raise HipError('the row store has been freed')
            radiance = np.zeros(S)
            npackets = np.zeros(S, dtype=np.int64)
            included = np.zeros(n_index, dtype=np.uint8) if n_index else None
            used = np.zeros((2, used_cap), dtype=np.int64) if used_cap else None
            n_used = C.c_int64(0)
            i64p = C.POINTER(C.c_int64)
            self._check(self.lib.nxc_los_accumulate_rows(
                self._h, C.byref(d), C.c_int64(S), _p(sc), store._r, C.c_int64(first),
                C.c_int64(count), C.c_int64(shift), C.c_int64(n_index), _p(radiance),
                npackets.ctypes.data_as(i64p),
                included.ctypes.data_as(C.POINTER(C.c_uint8)) if included is not None else None,
                C.c_int64(used_cap), used.ctypes.data_as(i64p) if used is not None else None,
                C.byref(n_used)))
            m = min(int(n_used.value), used_cap)
            return dict(radiance=radiance, npackets=npackets,
                        included=None if included is None else included.astype(bool),
                        used=None if used is None else used[:, :m], n_used=int(n_used.value))
        cols = (x, y, z, vy, frac)
        narrow = all(getattr(c, 'dtype', None) == np.float32 for c in cols)
        if narrow:          # stored float32 samples go over as they are; the device widens them
            cols = [np.ascontiguousarray(c) for c in cols]
            ptrs = [c.ctypes.data_as(C.POINTER(C.c_float)) for c in cols]
            entry = self.lib.nxc_los_accumulate_f32
        else:
            cols = [_f64(c) for c in cols]
            ptrs = [_p(c) for c in cols]
            entry = self.lib.nxc_los_accumulate
        P = len(cols[0])
        radiance = np.zeros(S)
        npackets = np.zeros(S, dtype=np.int64)
        idx = None if index is None else np.ascontiguousarray(index, dtype=np.int64)
        included = np.zeros(n_index, dtype=np.uint8) if n_index else None
        used = np.zeros((2, used_cap), dtype=np.int64) if used_cap else None
        n_used = C.c_int64(0)
        i64p = C.POINTER(C.c_int64)
        self._check(entry(
            self._h, C.byref(d), C.c_int64(S), _p(sc), C.c_int64(P), *ptrs,
            idx.ctypes.data_as(i64p) if idx is not None else None, C.c_int64(n_index),
            _p(radiance), npackets.ctypes.data_as(i64p),
            included.ctypes.data_as(C.POINTER(C.c_uint8)) if included is not None else None,
            C.c_int64(used_cap), used.ctypes.data_as(i64p) if used is not None else None,
            C.byref(n_used)))
        m = min(int(n_used.value), used_cap)
        return dict(radiance=radiance, npackets=npackets,
                    included=None if included is None else included.astype(bool),
                    used=None if used is None else used[:, :m], n_used=int(n_used.value))

    # -- RCCL -------------------------------------------------------------------------------
    def comm_unique_id(self):
        buf = (C.c_uint8*NXC_UNIQUE_ID_BYTES)()
        self._check(self.lib.nxc_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, nranks):
        buf = (C.c_uint8*NXC_UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self.lib.nxc_comm_init(self._h, buf, C.c_int(rank), C.c_int(nranks)))

    def comm_destroy(self):
        self._check(self.lib.nxc_comm_destroy(self._h))

    def image_allreduce(self):
        self._check(self.lib.nxc_image_allreduce(self._h))

    def allreduce_max(self, value):
        v = C.c_double(value)
        self._check(self.lib.nxc_allreduce_max_f64(self._h, C.byref(v)))
        return float(v.value)

    def allreduce_sum(self, value):
        v = C.c_double(value)
        self._check(self.lib.nxc_allreduce_sum_f64(self._h, C.byref(v)))
        return float(v.value)

    def barrier(self):
        self._check(self.lib.nxc_barrier(self._h))

    def allreduce(self, values):
        """Sum over the ranks of a small float64 array (returned; LOSResult.py:264-266 across
        GPUs)."""
        v = np.array(values, dtype=np.float64).ravel()
        self._check(self.lib.nxc_allreduce_f64(self._h, _p(v), C.c_int64(v.size)))
        return v.reshape(np.shape(values))

    def comm_set_timeout(self, seconds):
        """Deadline of every wait on a collective (default 120 s): past it the communicator is
        aborted and the waiting call raises HipError (code NXC_ERR_RCCL)."""
        self._check(self.lib.nxc_comm_set_timeout(self._h, C.c_double(seconds)))

    def comm_abort(self):
        self._check(self.lib.nxc_comm_abort(self._h))

    def comm_request_abort(self):
        """Thread-safe: ends the owning thread's wait on a collective (or its next collective)
        with HipError instead of letting it run to the deadline."""
        if self._h:
            self.lib.nxc_comm_request_abort(self._h)

    def comm_test_stall(self, seconds):
        """Fault injection: the stream is busy for ``seconds`` as if a collective hung."""
        self._check(self.lib.nxc_comm_test_stall(self._h, C.c_double(seconds)))

    # -- measurement helpers ----------------------------------------------------------------
    def stream_copy_gbs(self, nbytes=1 << 31, reps=5):
        """The box's streaming-copy rate, GB/s of bytes read + written."""
        gbs = C.c_double(0)
        self._check(self.lib.nxc_stream_copy_gbs(self._h, C.c_int64(nbytes), C.c_int(reps),
                                                 C.byref(gbs)))
        return float(gbs.value)

    def shader_clock_mhz(self):
        """The shader clock the chip holds under an fp64 load (in-kernel stamps)."""
        mhz = C.c_double(0)
        self._check(self.lib.nxc_shader_clock_mhz(self._h, C.byref(mhz)))
        return float(mhz.value)

    # -- diagnostics ------------------------------------------------------------------------
    def pcg64_uniforms(self, seed, n, row0, count, nvec):
        """(nvec, count): rows row0.. of the first nvec ``default_rng(seed).random(n)`` vectors,
        as the device sampler's generator 1 forms them."""
        state, inc = pcg64_words(seed)
        out = np.empty((nvec, count))
        self._check(self.lib.nxc_pcg64_uniforms(self._h, state, inc, C.c_int64(n), C.c_int64(row0),
                                                C.c_int64(count), C.c_int32(nvec), _p(out)))
        return out

    def math(self, which, x, y=None):
        code = {'exp': 0, 'log': 1, 'cube': 2, 'sqrt': 3, 'div': 4}[which]
        x = _f64(x)
        out = np.empty_like(x)
        y2 = _f64(y) if y is not None else None
        self._check(self.lib.nxc_math_batch(self._h, C.c_int(code), C.c_int64(len(x)), _p(x),
                                            _p(y2) if y2 is not None else None, _p(out)))
        return out
