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
  two-stage pipeline produces.

Bokeh display / PostgreSQL caching of the reference are out of scope.
"""
import copy
import json
import os

import numpy as np

from .atomicdata import gValue
from .input_classes import InputError
from .units import Quantity


def rotation_matrix(theta, axis):
    """Rotation by theta about axis (math/rotation_matrix.py:5-14)."""
    u = axis/np.linalg.norm(axis)
    lx, ly, lz = u[0], u[1], u[2]
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[lx**2+(1-lx**2)*c, lx*ly*(1-c)+lz*s, lx*lz*(1-c)-ly*s],
                     [lx*ly*(1-c)-lz*s, ly**2+(1-ly**2)*c, ly*lz*(1-c)+lx*s],
                     [lx*lz*(1-c)+ly*s, ly*lz*(1-c)-lx*s, lz**2+(1-lz**2)*c]])


class Histogram2dResult:
    """What math/histogram.py:28-39 exposes: histogram, bin centres x,y and widths dx,dy."""

    def __init__(self, histogram, xedges, yedges):
        self.histogram = histogram
        self.dx, self.dy = xedges[1]-xedges[0], yedges[1]-yedges[0]
        self.x = xedges[:-1] + self.dx/2
        self.y = yedges[:-1] + self.dy/2


class ModelResult:
    def __init__(self, inputs, params):
        self.inputs = copy.copy(inputs)
        self.outid, self.outputfiles, _, _ = inputs.search()
        self.npackets = 0
        self.totalsource = 0.
        self.atoms_per_packet = 0.
        self.sourcerate = Quantity(0., '1e23/s')
        if isinstance(params, str):
            if os.path.exists(params):
                self.params = {}
                with open(params, 'r') as f:
                    for line in f:
                        if ';' in line:
                            line = line[:line.find(';')]
                        elif '#' in line:
                            line = line[:line.find('#')]
                        if '=' in line:
                            p, v = line.split('=')
                            self.params[p.strip().lower()] = v.strip()
            else:
                raise FileNotFoundError('ModelResult.__init__', 'params file not found.')
        elif isinstance(params, dict):
            self.params = params
        else:
            raise TypeError('ModelResult.__init__', 'params must be a dict or filename.')

        quantities = ('column', 'radiance', 'density', 'difrad')
        self.quantity = self.params.get('quantity', None)
        if (self.quantity is None) or (self.quantity not in quantities):
            raise InputError('ModelImage.__init__', "quantity must be 'column' or 'radiance'")
        self.g = self.params.get('g', None)

        if self.quantity in ('radiance', 'difrad'):
            self.mechanism = ['resonant scattering']
            species = inputs.options.species
            if 'wavelength' in self.params:
                self.wavelength = tuple(sorted(int(m.strip()) for m
                                               in str(self.params['wavelength']).split(',')))
            elif species is None:
                raise InputError('ModelImage.__init__',
                                 'Must provide either species or params.wavelength')
            elif species == 'Na':
                self.wavelength = (5891, 5897)
            elif species == 'Ca':
                self.wavelength = (4227,)
            elif species == 'Mg':
                self.wavelength = (2852,)
            else:
                raise InputError('ModelResult.__init__',
                                 f'Default wavelengths not available for {species}')
        else:
            self.mechanism = None
            self.wavelength = None
        self.unit = 'R_' + inputs.geometry.planet.object
        self.unit_km = inputs.geometry.planet.radius.value

    def g_tables(self, aplanet):
        """[(velocity [R/s], g [1/s])] per emission line (ModelResult.py:152-157: gValue tables
        with the velocity axis converted to the packets' unit)."""
        if self.quantity not in ('radiance', 'difrad'):
            return []
        if self.g is not None:
            # constant g: a flat two-point table reproduces gg = g for every packet
            return [(np.array([-1e30, 1e30]), np.array([float(self.g), float(self.g)]))]
        tables = []
        for w in self.wavelength:
            gval = gValue(self.inputs.options.species, w, aplanet)
            tables.append((gval.velocity/self.unit_km, gval.g))
        return tables


class ModelImage(ModelResult):
    def __init__(self, inputs, params, overwrite=False, distribute=None, *, npackets=None,
                 seed=None, packs_per_it=None, downcast=True, device=0, context=None,
                 sampler='numpy', shard=None, finalize=True):
        super().__init__(inputs, params)
        self.type = 'image'
        self.origin = self.params.get('origin', inputs.geometry.planet)
        if self.origin != inputs.geometry.planet:
            raise NotImplementedError('images centred on another object '
                                      '(ModelResult.transform_reference_frame) are out of scope')

        dimtemp = str(self.params.get('dims', '800,800')).split(',')
        self.dims = [int(dimtemp[0]), int(dimtemp[1])]
        centtemp = str(self.params.get('center', '0,0')).split(',')
        self.center = [Quantity(float(centtemp[0]), self.unit),
                       Quantity(float(centtemp[1]), self.unit)]
        widtemp = str(self.params.get('width', '8,8')).split(',')
        self.width = [Quantity(float(widtemp[0]), self.unit),
                      Quantity(float(widtemp[1]), self.unit)]
        self.subobslongitude = Quantity(float(self.params.get('subobslongitude', '0')), 'rad')
        self.subobslatitude = Quantity(float(self.params.get('subobslatitude', np.pi/2)), 'rad')

        self.image = np.zeros(self.dims)
        self.packet_image = np.zeros(self.dims)
        self.blimits = None
        immin = tuple(c - w/2 for c, w in zip(self.center, self.width))
        immax = tuple(c + w/2 for c, w in zip(self.center, self.width))
        self.xrange = [immin[0], immax[0]]
        self.zrange = [immin[1], immax[1]]
        scale = tuple(w/d for w, d in zip(self.width, self.dims))
        R_cm = self.unit_km*1e5
        self.Apix = Quantity(scale[0]*scale[1]*R_cm**2, 'cm2')       # ModelImage.py:77-78
        self.xedges = np.linspace(self.xrange[0], self.xrange[1], self.dims[0]+1)
        self.zedges = np.linspace(self.zrange[0], self.zrange[1], self.dims[1]+1)
        self.xaxis = None
        self.zaxis = None
        self._ctx = context
        self._device = device
        self.counters = {}

        if npackets is not None:
            lo, hi = (0, int(npackets)) if shard is None else (int(shard[0]), int(shard[1]))
            if not 0 <= lo <= hi <= int(npackets):
                raise ValueError('shard must be an index range inside [0, npackets]')
            self._stream(int(npackets), seed, packs_per_it, downcast, sampler, lo, hi)
        else:
            outputs = [o for o in inputs._catalogue]
            if not outputs:
                print('No model outputs found for these inputs.')
            for out in outputs:
                print(f'Output filename: {out.filename}')
                image, packets = self.create_image(out)
                self.image += image.histogram
                self.packet_image += packets.histogram
                self.totalsource += out.totalsource
                self.xaxis = image.x
                self.zaxis = image.y

        if finalize:
            self.finalize()

    def finalize(self):
        """Scale to a source rate of 1e23 atoms/s (ModelImage.py:102-105); deferred by the
        multi-GPU path until the shards are summed."""
        mod_rate = self.totalsource / self.inputs.options.endtime.value
        self.atoms_per_packet = 1e23 / mod_rate if mod_rate > 0 else 0.
        self.sourcerate = Quantity(1., '1e23/s')
        self.image *= self.atoms_per_packet

    # ---- GPU plumbing ---------------------------------------------------------------------
    def context(self):
        if self._ctx is None:
            from . import hip_api
            self._ctx = hip_api.Context(self._device)
        return self._ctx

    def image_rotation(self):
        """ModelImage.py:367-384."""
        slong, slat = float(self.subobslongitude), float(self.subobslatitude)
        pSun = np.array([0., -1., 0.])
        pObs = np.array([np.sin(slong)*np.cos(slat), -np.cos(slong)*np.cos(slat), np.sin(slat)])
        if np.array_equal(pSun, pObs):
            return np.eye(3)
        costh = np.dot(pSun, pObs)/np.linalg.norm(pSun)/np.linalg.norm(pObs)
        theta = np.arccos(np.clip(costh, -1, 1))
        return rotation_matrix(theta, np.cross(pSun, pObs))

    def _set_image(self, ctx, aplanet, vrplanet_Rs, downcast):
        ctx.set_image(self.image_rotation(), vrplanet_Rs, float(self.Apix), self.quantity,
                      self.xedges, self.zedges, self.g_tables(aplanet), downcast_f32=downcast)

    def create_image(self, output):
        """ModelImage.py:229-274 for one catalogued Output (or .npz path): restore, rotate, mask,
        weight and histogram -- the last four inside one HIP kernel."""
        from .Output import Output
        output = Output.restore(output)
        packets = output.X
        if len(packets) == 0 or 'x' not in packets:
            raise ValueError('this Output holds no trajectory (it was run with '
                             'keep_trajectory=False); use ModelImage(..., npackets=N) instead')
        ctx = self.context()
        vr = float(output.vrplanet)/self.unit_km          # km/s -> R/s (ModelImage.py:242-243)
        self._set_image(ctx, float(output.aplanet), vr, downcast=False)
        ctx.image_accumulate(packets['x'].values, packets['y'].values, packets['z'].values,
                             packets['vy'].values, packets['frac'].values)
        self.counters = ctx.counters()
        assert self.counters['nonfinite'] == 0, 'Non-finite weights'
        image, counts = ctx.image_download()
        return (Histogram2dResult(image, self.xedges, self.zedges),
                Histogram2dResult(counts.astype(float), self.xedges, self.zedges))

    def _stream(self, total, seed, packs_per_it, downcast, sampler='numpy', lo=0, hi=None):
        """Fused integrate + image over the packets [lo, hi) of a run of ``total`` packets.

        The run is cut into chunks like Input.run does (Input.py:243-246); the chunk grid depends
        only on ``total`` and ``packs_per_it`` (distributed.chunk_plan), and chunk k of the host
        sampler is drawn from the generator seeded ``seed + k``, so the packets with global index
        in [lo, hi) are the same packets whether this process handles the whole run or one shard
        of it.  The device sampler is counter-based on the global index."""
        from .Output import Output, n_output_steps
        from .distributed import chunk_plan
        inputs = self.inputs
        opt = inputs.options
        if opt.step_size == 0:
            raise NotImplementedError('streaming images need constant-step inputs; the '
                                      'variable-step driver keeps one final row per packet')
        hi = total if hi is None else hi
        ctx = self.context()
        chunk = int(packs_per_it) if packs_per_it else max(1, min(total, 20_000_000))
        nsteps, n_iter = n_output_steps(opt.endtime.value, float(opt.step_size))
        first = True
        totals = {}
        src = bounce = bodies = None
        for k, c0, clen, a, b in chunk_plan(total, chunk, lo, hi):
            n = b - a
            if sampler == 'device' and not first:
                # same inputs, next slice of the counter space: no need to rebuild the tables
                ctx.sample_packets(n, 0 if seed is None else seed, a, **src)
            elif sampler == 'device':     # one counter space: packet i is draw block i
                out = Output(inputs, n, seed=seed, integrate=False, save=False, context=ctx,
                             sampler='device', first_index=a, materialize_x0=False)
                src, bounce, bodies = out.source_desc(), out._bounce, out._bodies
            else:
                # the whole chunk is drawn (the generator is sequential), rows [a, b) are kept
                out = Output(inputs, clen, seed=None if seed is None else seed + k,
                             integrate=False, save=False, context=ctx)
                bounce, bodies = out._bounce, out._bodies
            if first:
                ctx.set_forces(**out.forces_kwargs())
                self._set_image(ctx, out.aplanet, out.vrplanet, downcast)   # clears the image
                first = False
            if sampler != 'device':
                soa = out.x0_soa()
                ctx.upload_soa(soa if n == clen else np.ascontiguousarray(soa[:, a-c0:b-c0]))
            ctx.set_bounce(bounce)
            ctx.set_bodies(bodies)
            ctx.set_first_index(a)
            ctx.integrate_const(float(opt.step_size), n_iter, opt.outeredge, image=True)
            for key, v in ctx.counters().items():
                totals[key] = totals.get(key, 0) + v
            self.totalsource += n * nsteps                                  # Output.py:434
            self.npackets += n
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
        if filename.endswith('.json'):
            with open(filename, 'w') as f:
                json.dump({'image': self.image.tolist(), 'xaxis': self.xaxis.tolist(),
                           'zaxis': self.zaxis.tolist()}, f)
        else:
            raise TypeError('Not an valid file format')
