"""ModelResult / ModelImage: column-density and radiance images, accumulated on the GPU.

Drop-in for data_simulation/ModelResult.py:10-170 and data_simulation/ModelImage.py:26-105,
229-274,367-384 of the reference: ``ModelImage(inputs, params, overwrite=False, distribute=None)``
with ``params`` a dict or a ``key = value`` file (quantity, dims, center, width, subobslongitude,
subobslatitude, wavelength, g) and the attributes image, packet_image, xaxis, zaxis, Apix,
totalsource, atoms_per_packet, sourcerate, dims, center, width.

Two ways to get the packets:
* catalogued Outputs that hold trajectories (``inputs.run(...)``): each one is restored
  (float32 -> float64, Output.py:555-570) and binned by the HIP image kernel -- the reference's
  create_image data flow;
* STREAMING (keyword ``npackets``): integrate and bin in one persistent kernel; the trajectory
  tensor is never built.  ``downcast=True`` (default) applies the reference's float32
  save/restore rounding to every sample before binning, so the image is the one the reference's
  two-stage pipeline produces.  ``shard=(lo, hi)`` restricts the run to those global packet
  indices (multi-GPU, nexoclom_amd.distributed).

Bokeh display / PostgreSQL caching of the reference are out of scope.
"""
import copy
import json
import os

import numpy as np

from .atomicdata import gValue
from .input_classes import InputError
from .units import Quantity

HOST_SAMPLER_THREADS = 4        # chunks drawn ahead of the device by the streaming image
QUANTITIES = ('column', 'radiance', 'density', 'difrad')
EMISSION = ('radiance', 'difrad')
# resonance lines [Angstrom] summed when params name none (ModelResult.py:124-129)
DEFAULT_LINES = {'Na': (5891, 5897), 'Ca': (4227,), 'Mg': (2852,)}


def rotation_matrix(theta, axis):
    """Rotation by theta about axis, element for element math/rotation_matrix.py:5-14 (the image
    must bin with the same 3 x 3 doubles as the reference)."""
    u = axis/np.linalg.norm(axis)
    lx, ly, lz = u[0], u[1], u[2]
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[lx**2+(1-lx**2)*c, lx*ly*(1-c)+lz*s, lx*lz*(1-c)-ly*s],
                     [lx*ly*(1-c)-lz*s, ly**2+(1-ly**2)*c, ly*lz*(1-c)+lx*s],
                     [lx*lz*(1-c)+ly*s, ly*lz*(1-c)-lx*s, lz**2+(1-lz**2)*c]])


def read_params(params):
    """The format description of a model result: a dict as given, or the ``key = value`` lines of
    a file (``;`` / ``#`` comments, keys lower-cased; ModelResult.py:77-101)."""
    if isinstance(params, dict):
        return params
    if not isinstance(params, str):
        raise TypeError('ModelResult.__init__', 'params must be a dict or filename.')
    if not os.path.exists(params):
        raise FileNotFoundError('ModelResult.__init__', 'params file not found.')
    table = {}
    with open(params, 'r') as handle:
        for raw in handle:
            text = raw.split(';' if ';' in raw else '#', 1)[0]
            key, eq, value = text.partition('=')
            if eq:
                table[key.strip().lower()] = value.strip()
    return table


def _two(text, convert):
    first, second = str(text).split(',')[:2]
    return [convert(first), convert(second)]


class Histogram2dResult:
    """What math/histogram.py:28-39 exposes: histogram, bin centres x,y and widths dx,dy."""

    def __init__(self, histogram, xedges, yedges):
        self.histogram = histogram
        self.dx, self.dy = xedges[1]-xedges[0], yedges[1]-yedges[0]
        self.x = xedges[:-1] + self.dx/2
        self.y = yedges[:-1] + self.dy/2


class ModelResult:
    """What an image and a line-of-sight result share: the parsed ``params``, the observed
    quantity and, for emission, which lines' g-values weight the packets."""

    def __init__(self, inputs, params):
        self.inputs = copy.copy(inputs)
        self.outid, self.outputfiles, _, _ = inputs.search()
        self.npackets = 0
        self.totalsource = 0.
        self.atoms_per_packet = 0.
        self.sourcerate = Quantity(0., '1e23/s')
        self.params = read_params(params)
        self.quantity = self.params.get('quantity')
        if self.quantity not in QUANTITIES:
            raise InputError('ModelImage.__init__', "quantity must be 'column' or 'radiance'")
        self.g = self.params.get('g', None)
        emits = self.quantity in EMISSION
        self.mechanism = ['resonant scattering'] if emits else None
        self.wavelength = self._lines(inputs.options.species) if emits else None
        self.unit = 'R_' + inputs.geometry.planet.object
        self.unit_km = inputs.geometry.planet.radius.value

    def _lines(self, species):
        if 'wavelength' in self.params:
            return tuple(sorted(int(w.strip()) for w in str(self.params['wavelength']).split(',')))
        if species is None:
            raise InputError('ModelImage.__init__',
                             'Must provide either species or params.wavelength')
        if species not in DEFAULT_LINES:
            raise InputError('ModelResult.__init__',
                             f'Default wavelengths not available for {species}')
        return DEFAULT_LINES[species]

    def g_tables(self, aplanet):
        """[(velocity [R/s], g [1/s])] per emission line (ModelResult.py:152-157: gValue tables
        with the velocity axis converted to the packets' unit); a constant ``g`` in params is a
        flat two-point table, which makes every packet's g that constant."""
        if self.quantity not in EMISSION:
            return []
        if self.g is not None:
            flat = float(self.g)
            return [(np.array([-1e30, 1e30]), np.array([flat, flat]))]
        species = self.inputs.options.species
        lines = (gValue(species, w, aplanet) for w in self.wavelength)
        return [(line.velocity/self.unit_km, line.g) for line in lines]


class ModelImage(ModelResult):
    def __init__(self, inputs, params, overwrite=False, distribute=None, *, npackets=None,
                 seed=None, packs_per_it=None, downcast=True, device=0, context=None,
                 sampler='numpy', shard=None, finalize=True, generator='philox'):
        super().__init__(inputs, params)
        self.type = 'image'
        self.origin = self.params.get('origin', inputs.geometry.planet)
        if self.origin != inputs.geometry.planet:
            raise NotImplementedError('images centred on another object '
                                      '(ModelResult.transform_reference_frame) are out of scope')
        self._frame()
        self.image = np.zeros(self.dims)
        self.packet_image = np.zeros(self.dims)
        self.blimits = None
        self.xaxis = None
        self.zaxis = None
        self._ctx = context
        self._device = device
        self.counters = {}
        if npackets is not None:
            total = int(npackets)
            lo, hi = (0, total) if shard is None else (int(shard[0]), int(shard[1]))
            if not 0 <= lo <= hi <= total:
                raise ValueError('shard must be an index range inside [0, npackets]')
            self._stream(total, seed, packs_per_it, downcast, sampler, lo, hi, generator)
        else:
            self._from_catalogue()
        if finalize:
            self.finalize()

    def _frame(self):
        """Image plane from params (ModelImage.py:53-78): dims (default 800 x 800), center and
        width in planet radii (0,0 and 8 x 8), the sub-observer point (longitude 0, latitude
        pi/2: looking down on the north pole), the bin edges and the pixel area in cm^2."""
        get = self.params.get
        length = lambda text: Quantity(float(text), self.unit)          # noqa: E731
        self.dims = _two(get('dims', '800,800'), int)
        self.center = _two(get('center', '0,0'), length)
        self.width = _two(get('width', '8,8'), length)
        self.subobslongitude = Quantity(float(get('subobslongitude', '0')), 'rad')
        self.subobslatitude = Quantity(float(get('subobslatitude', np.pi/2)), 'rad')
        self.xrange, self.zrange = ([c - w/2, c + w/2] for c, w in zip(self.center, self.width))
        pixel = [w/d for w, d in zip(self.width, self.dims)]
        R_cm = self.unit_km*1e5
        self.Apix = Quantity(pixel[0]*pixel[1]*R_cm**2, 'cm2')
        self.xedges = np.linspace(self.xrange[0], self.xrange[1], self.dims[0]+1)
        self.zedges = np.linspace(self.zrange[0], self.zrange[1], self.dims[1]+1)

    def _from_catalogue(self):
        """Sum of the images of every catalogued Output (ModelImage.py:85-98)."""
        runs = list(self.inputs._catalogue)
        if not runs:
            print('No model outputs found for these inputs.')
        if self._from_resident(runs):
            return
        for run in runs:
            print(f'Output filename: {run.filename}')
            weighted, counted = self.create_image(run)
            self.image += weighted.histogram
            self.packet_image += counted.histogram
            self.totalsource += run.totalsource
            self.xaxis, self.zaxis = weighted.x, weighted.y

    def _from_resident(self, runs):
        """The same sum when every run's rows are still in HBM and the runs share their geometry
        (one Input.run): the image pair stays on the device across the runs -- it IS the running
        sum of ModelImage.py:96-98 -- and comes to the host once.  False: not applicable."""
        from .Output import Output
        if not runs or not all(isinstance(run, Output) for run in runs):
            return False
        ctx = self.context()
        views = [run.resident_rows(ctx) for run in runs]
        same = {(float(run.aplanet), float(run.vrplanet)) for run in runs}
        if any(v is None for v in views) or len(same) != 1:
            return False
        (aplanet, vrplanet_kms), = same
        self._set_image(ctx, aplanet, vrplanet_kms/self.unit_km, downcast=False)    # clears it
        totals = {}

        def accumulate(store, first, count):
            if count:
                ctx.image_accumulate_rows(store, first, count)
                for key, v in ctx.counters().items():
                    totals[key] = totals.get(key, 0) + v

        # the Outputs of a launch group are consecutive slices of one store: one kernel launch per
        # run of adjacent slices instead of one per Output
        span = None                               # (store, first row, row count)
        for run, (store, first, count, _) in zip(runs, views):
            print(f'Output filename: {run.filename}')
            if span is not None and span[0] is store and span[1] + span[2] == first:
                span = (store, span[1], span[2] + count)
            else:
                if span is not None:
                    accumulate(*span)
                span = (store, first, count)
            self.totalsource += run.totalsource
        if span is not None:
            accumulate(*span)
        self.counters = totals
        assert totals.get('nonfinite', 0) == 0, 'Non-finite weights'
        image, counts = ctx.image_download()
        self.image += image
        self.packet_image += counts.astype(float)
        h = Histogram2dResult(image, self.xedges, self.zedges)
        self.xaxis, self.zaxis = h.x, h.y
        return True

    def finalize(self):
        """Scale to a source rate of 1e23 atoms/s (ModelImage.py:102-105); deferred by the
        multi-GPU path until the shards are summed."""
        per_second = self.totalsource / self.inputs.options.endtime.value
        self.atoms_per_packet = 1e23 / per_second if per_second > 0 else 0.
        self.sourcerate = Quantity(1., '1e23/s')
        self.image *= self.atoms_per_packet

    # ---- GPU plumbing ---------------------------------------------------------------------
    def context(self):
        if self._ctx is None:
            # the device the catalogued runs were made on, when there is one (a new handle costs
            # 0.1 s); else a fresh one
            shared = [getattr(run, '_ctx', None) for run in getattr(self.inputs, '_catalogue', ())]
            shared = [ctx for ctx in shared if ctx is not None and getattr(ctx, '_h', True)]
            if shared:
                self._ctx = shared[-1]
            else:
                from . import hip_api
                self._ctx = hip_api.Context(self._device)
        return self._ctx

    def image_rotation(self):
        """Sun frame -> observer frame (ModelImage.py:367-384): the rotation that carries the
        sub-solar direction (0,-1,0) onto the sub-observer direction."""
        lon, lat = float(self.subobslongitude), float(self.subobslatitude)
        sun = np.array([0., -1., 0.])
        observer = np.array([np.sin(lon)*np.cos(lat), -np.cos(lon)*np.cos(lat), np.sin(lat)])
        if np.array_equal(sun, observer):
            return np.eye(3)
        cosine = np.dot(sun, observer)/np.linalg.norm(sun)/np.linalg.norm(observer)
        return rotation_matrix(np.arccos(np.clip(cosine, -1, 1)), np.cross(sun, observer))

    def _set_image(self, ctx, aplanet, vrplanet_Rs, downcast):
        ctx.set_image(self.image_rotation(), vrplanet_Rs, float(self.Apix), self.quantity,
                      self.xedges, self.zedges, self.g_tables(aplanet), downcast_f32=downcast)

    def create_image(self, output):
        """ModelImage.py:229-274 for one catalogued Output (or .npz path): the restored sample
        columns are rotated, masked, weighted and binned inside one HIP kernel."""
        from .Output import Output
        ctx = self.context()
        view = output.resident_rows(ctx) if isinstance(output, Output) else None
        if view is not None and view[2] > 0:
            # the rows are still in HBM as save() would have stored them: bin them where they are
            aplanet, vrplanet_kms = float(output.aplanet), float(output.vrplanet)
            self._set_image(ctx, aplanet, vrplanet_kms/self.unit_km, downcast=False)
            ctx.image_accumulate_rows(view[0], view[1], view[2])
        else:
            samples, aplanet, vrplanet_kms = Output.image_columns(output)
            if samples is None or len(samples[0]) == 0:
                raise ValueError('this Output holds no trajectory (it was run with '
                                 'keep_trajectory=False); use ModelImage(..., npackets=N) instead')
            vr = vrplanet_kms/self.unit_km                 # km/s -> R/s (ModelImage.py:242-243)
            self._set_image(ctx, aplanet, vr, downcast=False)
            ctx.image_accumulate(*samples)
        self.counters = ctx.counters()
        assert self.counters['nonfinite'] == 0, 'Non-finite weights'
        image, counts = ctx.image_download()
        return (Histogram2dResult(image, self.xedges, self.zedges),
                Histogram2dResult(counts.astype(float), self.xedges, self.zedges))

    def _stream(self, total, seed, packs_per_it, downcast, sampler='numpy', lo=0, hi=None,
                generator='philox'):
        """Fused integrate + image over the packets [lo, hi) of a run of ``total`` packets.

        The run is cut into chunks like Input.run does (Input.py:243-246); the chunk grid depends
        only on ``total`` and ``packs_per_it`` (distributed.chunk_plan), and chunk k of the host
        sampler is drawn from the generator seeded ``seed + k``, so the packets with global index
        in [lo, hi) are the same packets whether this process handles the whole run or one shard
        of it.  The device sampler is counter-based on the global index
        (generator='philox'), or follows the host sampler's own seeded streams, chunk k from
        ``seed + k`` (generator='pcg64': the same packets as sampler='numpy' to libm rounding,
        drawn where they are integrated)."""
        from .Output import Output, n_output_steps
        from .distributed import chunk_plan
        inputs = self.inputs
        opt = inputs.options
        if opt.step_size == 0:
            raise NotImplementedError('streaming images need constant-step inputs; the '
                                      'variable-step driver keeps one final row per packet')
        hi = total if hi is None else hi
        ctx = self.context()
        if seed is None and sampler == 'device':
            # an unseeded run: one fresh key for all its chunks (recorded; default_rng(None) of
            # the host sampler is entropy-seeded too), not the same key 0 for every run
            from .Output import fresh_key
            seed = fresh_key()
        self.seed = seed
        chunk = int(packs_per_it) if packs_per_it else max(1, min(total, 20_000_000))
        nsteps, n_iter = n_output_steps(opt.endtime.value, float(opt.step_size))
        first = True
        totals = {}
        src = bounce = bodies = None
        plan = list(chunk_plan(total, chunk, lo, hi))

        from .source_distribution import WindowGenerator
        windowed = seed is not None and WindowGenerator.windowable(inputs)

        def host_chunk(k, clen, lo_row, hi_row):
            # a rank that owns only rows [lo_row, hi_row) of the chunk draws only those (the
            # seeded stream is jumped into with PCG64.advance: bit-identical to slicing the whole
            # draw); sources whose draws cannot be windowed draw the chunk and slice
            if windowed and (lo_row, hi_row) != (0, clen):
                return Output(inputs, clen, seed=seed + k, window=(clen, lo_row, hi_row),
                              integrate=False, save=False, context=ctx)
            return Output(inputs, clen, seed=None if seed is None else seed + k,
                          integrate=False, save=False, context=ctx)

        # host-sampled chunks are independent generators (seed + k): the next ones are drawn by
        # worker threads (NumPy releases the GIL) while the device integrates the current one
        ahead, pool = {}, None
        if sampler != 'device' and len(plan) > 1:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=min(len(plan), HOST_SAMPLER_THREADS))
        try:
            for position, (k, c0, clen, a, b) in enumerate(plan):
                n = b - a
                if sampler == 'device' and generator == 'pcg64':
                    if seed is None:
                        raise ValueError("generator='pcg64' needs a seed")
                    if first:
                        out = Output(inputs, clen, seed=seed + k, integrate=False, save=False,
                                     context=ctx, sampler='device', generator='pcg64',
                                     window=(clen, a - c0, b - c0), first_index=a,
                                     materialize_x0=False)
                        src, bounce, bodies = out.source_desc(), out._bounce, out._bodies
                    else:
                        ctx.sample_packets(n, seed + k, a, pcg64=(clen, a - c0), **src)
                elif sampler == 'device' and not first:
                    # same inputs, next slice of the counter space: no need to rebuild the tables
                    ctx.sample_packets(n, 0 if seed is None else seed, a, **src)
                elif sampler == 'device':     # one counter space: packet i is draw block i
                    out = Output(inputs, n, seed=seed, integrate=False, save=False, context=ctx,
                                 sampler='device', first_index=a, materialize_x0=False)
                    src, bounce, bodies = out.source_desc(), out._bounce, out._bodies
                else:
                    if pool is not None:
                        for later in plan[position:position + HOST_SAMPLER_THREADS]:
                            if later[0] not in ahead:
                                ahead[later[0]] = pool.submit(host_chunk, later[0], later[2],
                                                              later[3] - later[1],
                                                              later[4] - later[1])
                        out = ahead.pop(k).result()
                    else:
                        out = host_chunk(k, clen, a - c0, b - c0)
                    bounce, bodies = out._bounce, out._bodies
                if first:
                    ctx.set_forces(**out.forces_kwargs())
                    self._set_image(ctx, out.aplanet, out.vrplanet, downcast)   # clears the image
                    first = False
                if sampler != 'device':
                    soa = out.x0_soa()
                    ctx.upload_soa(soa if soa.shape[1] == n
                                   else np.ascontiguousarray(soa[:, a-c0:b-c0]))
                ctx.set_bounce(bounce)
                ctx.set_bodies(bodies)
                ctx.set_first_index(a)
                ctx.integrate_const(float(opt.step_size), n_iter, opt.outeredge, image=True)
                for key, v in ctx.counters().items():
                    totals[key] = totals.get(key, 0) + v
                self.totalsource += n * nsteps                                  # Output.py:434
                self.npackets += n
        finally:
            if pool is not None:
                pool.shutdown(cancel_futures=True)
        self.counters = totals
        assert totals.get('nonfinite', 0) == 0, 'Non-finite weights'
        if first:       # an empty shard still owns a resident (zero) image for the reduce
            out = Output(inputs, 0, seed=seed, integrate=False, save=False, context=ctx)
            ctx.set_forces(**out.forces_kwargs())
            self._set_image(ctx, out.aplanet, out.vrplanet, downcast)
        image, counts = ctx.image_download()
        self.image += image
        self.packet_image += counts.astype(float)
        h = Histogram2dResult(image, self.xedges, self.zedges)
        self.xaxis, self.zaxis = h.x, h.y

    def export(self, filename='image.json'):
        if not filename.endswith('.json'):
            raise TypeError('Not an valid file format')
        with open(filename, 'w') as handle:
            json.dump({k: getattr(self, k).tolist() for k in ('image', 'xaxis', 'zaxis')}, handle)
