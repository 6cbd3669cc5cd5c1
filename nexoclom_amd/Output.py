"""Output: packet trajectories, integrated on the GPU.

Drop-in for the reference's particle_tracking/Output.py:23-455 on the particle-tracking path:
same constructor signature ``Output(inputs, npackets, compress=True, run_model=True, seed=None)``,
same attributes (X0, X, npackets, totalsource, nsteps, unit, GM, aplanet, vrplanet, radpres,
loss_info, randgen, compress, inputs, planet) and the same column order of ``X``.  What differs:

* ``constant_step_size_driver`` / ``variable_step_size_driver`` call the HIP kernels through the
  C ABI (nexoclom_amd.hip_api) instead of looping over NumPy rk5 steps.  There is NO CPU
  integrator in this package: without the HIP library or a GPU these methods raise.
* ``save()`` writes an .npz next to an in-memory catalogue instead of PostgreSQL + pickle
  (persistence is out of scope), but applies the same ``compress`` row filter
  and float32 down-cast (Output.py:522-543) so that what ModelImage later reads is what the
  reference would read.
* keyword-only extras: ``device`` (GPU index), ``integrate`` (False: set up and sample X0 only),
  ``keep_trajectory`` (False: constant-step runs never materialise the (N, 8, nsteps) tensor --
  used by the fused integrate+image path), ``context`` (share one hip_api.Context),
  ``sampler`` ('numpy': the reference's seeded draw order on the host; 'device': Philox on the
  GPU, statistically equivalent, for runs where host sampling would dominate),
  ``first_index`` (offset of this chunk in the device sampler's counter space),
  ``materialize_x0`` (False: leave the device-sampled states on the GPU), ``window``
  ((n, a, b): only rows [a, b) of the n packets the seed would draw, bit-identical to slicing)
  and ``generator`` ('pcg64': the device sampler follows the seeded HOST stream -- NumPy's PCG64,
  the uniforms bit for bit -- instead of its own Philox counters; uniform / flat / isotropic |
  radial sources).
* the trajectory rows of a constant-step run stay in HBM (hip_api.RowStore) in exactly the form
  save() would store them -- frac > 0 rows, float32 / int32 -- and ``X`` is built from them on
  first access; ModelImage and LOSResult read the resident rows directly.  ``Output.integrate_batch``
  integrates several Outputs of one ``Input.run`` in a single launch.
"""
import os

import numpy as np
import pandas as pd

from .atomicdata import LossInfo, RadPresConst
from .solarsystem import planet_dist
from .source_distribution import (LaunchTable, angular_distribution, speed_distribution,
                                  surface_distribution)
from .units import Quantity, register_unit

STATE_COLS = ['time', 'x', 'y', 'z', 'vx', 'vy', 'vz', 'frac']
# save()'s 32-bit down-cast and restore()'s way back (Output.py:528-543, 555-570)
NARROW = {np.int64: np.int32, np.float64: np.float32}
WIDE = {np.int32: np.int64, np.float32: np.float64}


def fresh_key():
    """A 64-bit key for the counter-based device generators of an UNSEEDED run, from the
    operating system's entropy like numpy.random.default_rng(None) (Output.py:92): two unseeded
    runs must not integrate identical packets."""
    return int(np.random.SeedSequence().entropy) & 0xffffffffffffffff


def n_output_steps(endtime, step):
    """nsteps (Output.py:375) and the number of iterations of the ``while curtime > 0`` loop
    (Output.py:384,431), evaluated with the same float arithmetic."""
    nsteps = int(np.ceil(endtime/step + 1))
    curtime, iters = float(endtime), 0
    while curtime > 0:
        iters += 1
        curtime -= step
    return nsteps, min(iters, nsteps-1)


class Output:
    # trajectory rows resident on the device (a view of a hip_api.RowStore) until X is asked for
    _X = None
    _store = None
    _row0 = _nrows = _packet0 = 0
    _lengths = None

    def __init__(self, inputs, npackets, compress=True, run_model=True, seed=None, *,
                 device=0, integrate=True, keep_trajectory=True, context=None, save=True,
                 sampler='numpy', first_index=0, materialize_x0=True, presampled=False,
                 window=None, generator='philox'):
        self.inputs = inputs
        self.planet = inputs.geometry.planet
        # a finished reference Output always went through save() (Output.py:202): its frames are
        # 32-bit.  When this one will too, the rows can leave the device already narrowed.
        self._narrow_rows = bool(save and run_model)
        self._ctx = context
        self._device = device
        self.filename = None
        self.idnum = None
        if run_model:
            if window is None:
                self.randgen = np.random.default_rng(seed=seed)
            else:
                # rows [a, b) of the npackets = window[0] packets the seed would draw: only those
                # are drawn (multi-GPU shards, ModelImage._stream); everything below sees b - a
                from .source_distribution import WindowGenerator
                self.randgen = WindowGenerator(seed, *window) if sampler == 'numpy' else None
                npackets = window[2] - window[1]
            if generator not in ('philox', 'pcg64'):
                raise ValueError("generator must be 'philox' or 'pcg64'")
            if generator == 'pcg64' and (sampler != 'device' or seed is None):
                raise ValueError("generator='pcg64' is the device sampler's reproduction of the "
                                 "seeded host stream: it needs sampler='device' and a seed")
            self.generator = generator
            # the parser calls it 'geometry with starttime' (input_classes.py:75), which slips past
            # the reference's assert and dies on the missing .taa; both spellings stop here
            assert self.inputs.geometry.type not in ('geometry with time',
                                                     'geometry with starttime'), (
                'Initialization with time stamp not implemented yet.')
            self.compress = compress
            npackets = int(npackets)

            # length unit = planet radius (Output.py:102)
            self.unit = 'R_' + self.planet.object
            self.unit_km = self.planet.radius.value
            register_unit(self.unit, 'length', self.unit_km*1e3)
            # GM in R^3/s^2, negative (Output.py:105; SSObject.py:53)
            self.GM = self.planet.GM.value / (self.unit_km*1e3)**3

            r, v_r = planet_dist(self.planet, inputs.geometry.taa)      # Output.py:108-110
            self.aplanet = r.value
            self.vrplanet = v_r.value / self.unit_km

            if inputs.options.lifetime.value <= 0:                      # Output.py:113-118
                self.loss_info = LossInfo(inputs.options.species, inputs.options.lifetime,
                                          self.aplanet)
            else:
                self.loss_info = None

            if inputs.forces.radpres:                                   # Output.py:121-128
                radpres = RadPresConst(inputs.options.species, self.aplanet)
                radpres.velocity = radpres.velocity / self.unit_km
                radpres.accel = radpres.accel / self.unit_km
                self.radpres = radpres
            else:
                self.radpres = None

            # surface accommodation / sticking set-up when packets do not simply stick
            # (Output.py:130-133); None = absorbed on impact
            # key of the device's counter-based draws (initial states, re-emission): the seed, or
            # for an unseeded run a fresh one, kept on the Output so that the run can be repeated
            self.device_key = fresh_key() if seed is None else int(seed)
            from .surface import bounce_config
            self._bounce = bounce_config(inputs, self.GM, self.unit_km, self.device_key)
            self._first_index = first_index

            self.npackets = npackets
            # the host sampler fills a table of plain columns; the frame is built once at the end
            self.X0 = LaunchTable(npackets) if sampler == 'numpy' else pd.DataFrame()
            if sampler == 'device':
                # the device draws everything, launch times included; host columns of npackets
                # rows (0.2 s per 2e7) are built only when the caller wants X0 back
                self.totalsource = float(npackets)
            else:
                if inputs.options.step_size != 0:                       # Output.py:136-141
                    time = np.ones(npackets) * inputs.options.endtime.value
                else:
                    time = self.randgen.random(npackets) * inputs.options.endtime.value
                self.X0['time'] = time
                self.X0['frac'] = np.ones(npackets)
                self.totalsource = self.X0['frac'].sum()

            # The reference stops here for any planet with moons ('Not set up',
            # Output.py:153-155).  EXTENSION: included moons pull on and absorb packets, a moon
            # can be the start point, and options.chx_* adds a plasma-torus loss term
            # (include/nexoclom_hip.h, nxc_bodies_desc).
            self._bodies = self._bodies_config()

            self.sampler = sampler
            self._resident = False
            if sampler == 'device':
                if inputs.geometry.planet.object != inputs.geometry.startpoint:
                    raise NotImplementedError("sampler='device' launches from the planet only")
                # presampled: the caller drew this Output's packets as part of a larger device call
                # (Input.run takes all its chunks in one go)
                soa = None if presampled else self.context().sample_packets(
                    npackets, self.device_key, first_index,
                    download=materialize_x0, **self.stream_window(seed, window),
                    **self.source_desc())
                self._resident = not presampled
                if materialize_x0 and not presampled:
                    self._adopt_x0(soa)
                else:
                    self.X0 = pd.DataFrame()
            elif sampler == 'numpy':
                if inputs.spatialdist.type in ('uniform', 'surface map', 'surface spot'):
                    surface_distribution(self)
                else:
                    assert 0, 'Not a valid spatial distribution type'
                speed_distribution(self)
                angular_distribution(self)
                self._launch_from_moon()
                cols = ['time', 'x', 'y', 'z', 'vx', 'vy', 'vz', 'frac', 'v',
                        'longitude', 'latitude', 'local_time', 'altitude', 'azimuth']
                self.X0 = self.X0.frame(cols)
            else:
                raise ValueError("sampler must be 'numpy' or 'device'")
            self.nsteps = None
            self.X = pd.DataFrame()
            self.counters = {}

            if integrate:
                if inputs.options.step_size == 0:
                    print('Running variable step size integrator.')
                    self.X = self.X0.drop(['longitude', 'latitude', 'local_time'], axis=1, errors='ignore')
                    self.X['lossfrac'] = np.zeros(npackets)
                    self.variable_step_size_driver()
                else:
                    print('Running constant step size integrator.')
                    self.constant_step_size_driver(keep_trajectory=keep_trajectory)
        else:
            print('Not running anything')
            self.compress = False
            self.X0 = pd.DataFrame()
            self.X = pd.DataFrame()
            self.npackets = npackets
            self.totalsource = npackets
        if save and (not run_model or integrate):
            self.save()

    def __len__(self):
        return self.npackets

    def __str__(self):
        return (f'Contents of output:\n\tPlanet = {self.planet.object}\n'
                f'\ta_planet = {self.aplanet}\n\tvr_planet = {self.vrplanet}\n'
                f'\tNumber of Packets: {self.npackets}')

    # ---- GPU plumbing ---------------------------------------------------------------------
    def context(self):
        if self._ctx is None:
            from . import hip_api
            self._ctx = hip_api.Context(self._device)
        return self._ctx

    def forces_kwargs(self):
        """Arguments of hip_api.Context.set_forces for this run (what state() reads off
        ``output``: state.py:19,27,34-36,44-52)."""
        opt = self.inputs.options
        photo = None
        if self.loss_info is not None and self.loss_info.photo is not None:
            photo = float(self.loss_info.photo)
        kw = dict(GM=float(self.GM), vrplanet=float(self.vrplanet),
                  gravity=bool(self.inputs.forces.gravity),
                  radpres=bool(self.inputs.forces.radpres),
                  lifetime=float(opt.lifetime.value), photo=photo)
        if self.radpres is not None:
            kw.update(v_tab=self.radpres.velocity, a_tab=self.radpres.accel)
        return kw

    def stream_window(self, seed, window=None):
        """``pcg64=`` argument of hip_api.Context.sample_packets for this Output: with
        generator='pcg64' its packets are rows [a, b) of the vectors default_rng(seed) would draw
        for ``window = (n, a, b)`` (the whole Output when there is no window); {} for Philox."""
        if getattr(self, 'generator', 'philox') != 'pcg64':
            return {}
        return {'pcg64': (self.npackets, 0) if window is None else (window[0], window[1])}

    def source_desc(self):
        """Keyword arguments of hip_api.Context.sample_packets for these inputs: the scalar fields
        of nxc_source_desc plus, where the source needs them, ``speed_table`` (maxwellian /
        sputtering: the inverse-CDF table the host sampler interpolates in,
        math/randomdeviates.py:29-33) and ``surface_map`` (surface spot: the 361 x 181 density map
        the host sampler rejects against, source_distribution.py:96-113)."""
        from .source_distribution import density_cdf, spot_density_map, tabulated_speed_density
        sd, vd, ad = self.inputs.spatialdist, self.inputs.speeddist, self.inputs.angulardist
        if sd.type not in ('uniform', 'surface spot') \
                or vd.type not in ('flat', 'gaussian', 'maxwellian', 'sputtering') \
                or ad.type not in ('isotropic', 'radial'):
            raise NotImplementedError("sampler='device' supports uniform|surface spot / "
                                      "flat|gaussian|maxwellian|sputtering / isotropic|radial "
                                      "sources")
        d = dict(endtime=self.inputs.options.endtime.value, exobase=float(sd.exobase),
                 unit_km=self.unit_km, random_time=int(self.inputs.options.step_size == 0),
                 angular_type=0 if ad.type == 'radial' else 1,
                 is_planet=int(self.planet.type == 'Planet'),
                 sinlat0=-1.0, sinlat1=1.0, lon0=0.0, lon1=2*np.pi, vprob=0.0, vwidth=0.0,
                 sinalt0=0.0, sinalt1=1.0, az0=0.0, az1=2*np.pi)
        if sd.type == 'uniform':
            lon0, lon1 = (float(v) for v in sd.longitude)
            d.update(spatial_type=0, lon0=lon0, lon1=lon1 + 2*np.pi if lon0 > lon1 else lon1,
                     sinlat0=float(np.sin(sd.latitude[0])), sinlat1=float(np.sin(sd.latitude[1])))
        else:
            _, _, density = spot_density_map(float(sd.longitude), float(sd.latitude),
                                             float(sd.sigma))
            d.update(spatial_type=1, surface_map=density)
        if vd.type in ('flat', 'gaussian'):
            d.update(speed_type=0 if vd.type == 'flat' else 1, vprob=vd.vprob.value,
                     vwidth=vd.delv.value if vd.type == 'flat' else vd.sigma.value)
        else:
            grid, density = tabulated_speed_density(vd, self.inputs.options.species)
            d.update(speed_type=2, speed_table=density_cdf(grid, density))
        if ad.type == 'isotropic':
            az0, az1 = (float(v) for v in ad.azimuth)
            if az0 > az1:
                az0, az1 = az1, az0 + 2*np.pi
            d.update(sinalt0=float(np.sin(ad.altitude[0])), sinalt1=float(np.sin(ad.altitude[1])),
                     az0=az0, az1=az1)
        return d

    def _bodies_config(self):
        """Arguments of hip_api.Context.set_bodies, or None for the reference's single-body
        model.  Moons are taken in the order of ``planet.moons``; geometry.phi follows it."""
        geo, opt = self.inputs.geometry, self.inputs.options
        included = [m for m in (self.planet.moons or []) if m in (geo.objects or ())]
        chx = getattr(opt, 'chx', None)
        if not included and chx is None:
            return None
        if opt.step_size == 0:
            raise NotImplementedError('moons / torus loss need the constant-step driver '
                                      '(options.step_size > 0)')
        unit_m = self.unit_km*1e3
        cfg = dict(moons=[], t0=float(opt.endtime.value), chx=None)
        phis = [float(p) for p in (getattr(geo, 'phi', None) or ())]
        assert len(phis) == len(included), 'The wrong number of orbital positions was given.'
        for m, phi in zip(included, phis):
            cfg['moons'].append(dict(name=m.object, gm=m.GM.value/unit_m**3,
                                     radius=m.radius.value/self.unit_km,
                                     a=m.a.value/self.unit_km,
                                     omega=2*np.pi/(m.orbperiod.value*86400.), phi=phi))
        if chx is not None:
            omega = 2*np.pi/(self.planet.rotperiod.value*3600.) if chx['corotation'] else 0.0
            cfg['chx'] = dict(k0=chx['k0'], rho0=chx['rho0'], width=chx['width'],
                              height=chx['height'], omega=omega)
        return cfg

    def _launch_from_moon(self):
        """Move packets sampled around the origin onto the start-point moon: scale to its radius,
        rotate its local frame (-y towards the planet, -x leading) by the orbital phase at launch,
        add its position, orbital velocity and the surface's co-rotation."""
        geo = self.inputs.geometry
        if geo.planet.object == geo.startpoint:
            return
        mo = next((m for m in self._bodies['moons'] if m['name'] == geo.startpoint), None)
        assert mo is not None, 'geometry.startpoint must be one of geometry.objects'
        X0 = self.X0
        ang = mo['phi'] - mo['omega']*X0['time']
        c, s_ = np.cos(ang), np.sin(ang)
        xl, yl = X0['x']*mo['radius'], X0['y']*mo['radius']
        xr, yr = c*xl - s_*yl, s_*xl + c*yl
        vxl, vyl = X0['vx'], X0['vy']
        aw = mo['a']*mo['omega']
        X0['x'] = xr - mo['a']*s_
        X0['y'] = yr + mo['a']*c
        X0['z'] = X0['z']*mo['radius']
        X0['vx'] = (c*vxl - s_*vyl) - aw*c - mo['omega']*yr
        X0['vy'] = (s_*vxl + c*vyl) - aw*s_ + mo['omega']*xr

    def _adopt_x0(self, soa):
        """X0 of a device-sampled Output from its (8, n) block of the downloaded states."""
        self.X0 = pd.DataFrame({c: soa[k] for k, c in enumerate(STATE_COLS)})
        self.X0['v'] = np.sqrt(self.X0.vx**2 + self.X0.vy**2 + self.X0.vz**2)

    def upload(self, ctx):
        """Make this Output's initial states the context's resident packet set."""
        if getattr(self, '_resident', False) and ctx is self._ctx:
            return
        ctx.upload_soa(self.x0_soa())

    def x0_soa(self):
        """Initial state as the (8, N) struct-of-arrays block the C ABI takes."""
        ready = getattr(self, '_soa_ready', None)
        if ready is not None:
            return ready
        return np.ascontiguousarray(self.X0[STATE_COLS].values.T, dtype=np.float64)

    def prepare_for_launch(self, will_save=True):
        """What the launch and save() need of X0, made ahead of time (Input.run calls this on the
        thread that drew the packets, beside the device): the (8, N) upload block and, for
        save(), the float32 frame of Output.py:528-543."""
        self._soa_ready = self.x0_soa()
        if will_save:
            self._x0_narrow = self._recast(self.X0, NARROW)
        return self

    def _raise_on_counters(self, ctr):
        self.counters = ctr
        assert ctr.get('nonfinite', 0) == 0, '\n\tInfinite values of emax'
        assert ctr.get('neg_frac', 0) == 0, 'Found new values of frac that are negative'
        assert ctr.get('bad_step', 0) == 0, 'Bad step size'

    # ---- X: built from the resident rows on first access ------------------------------------
    @property
    def X(self):
        if self._X is None and self._store is not None:
            self._X = self._frame_from_rows()
        return self._X

    @X.setter
    def X(self, frame):
        self._X = frame

    def _attach_rows(self, store, row0, lengths, packet0):
        """This Output's trajectories are rows [row0, row0 + sum(lengths)) of ``store``; its
        packets are numbers packet0 .. packet0 + npackets - 1 of the store's index column."""
        self._store, self._row0, self._packet0 = store, int(row0), int(packet0)
        self._lengths = lengths
        self._nrows = int(lengths.sum())
        self._X = None
        store.owners.add(self)

    def resident_rows(self, ctx=None):
        """(store, first row, row count, first packet) while the rows live in HBM (on ``ctx``'s
        device when given), else None."""
        store = self._store
        if store is None or store._r is None or (ctx is not None and store.ctx is not ctx):
            return None
        return store, self._row0, self._nrows, self._packet0

    def _host_rows(self):
        """(rows (9, n), Index (n,)) of this Output as save() stores them, from HBM."""
        rows, idx = self._store.download(self._row0, self._nrows)
        if self._packet0:
            idx -= idx.dtype.type(self._packet0)
        return rows, idx

    def _frame_from_rows(self):
        """The reference's X after save()'s frac > 0 filter (Output.py:435-449,523-524): the
        surviving rows keep their original labels packet*nsteps + ct."""
        rows, idx = self._host_rows()
        n, lengths = self.npackets, self._lengths
        starts = np.cumsum(lengths) - lengths
        labels = np.repeat(np.arange(n, dtype=np.int64)*self.nsteps - starts, lengths)
        labels += np.arange(len(labels), dtype=np.int64)
        columns = {'Index': idx}
        columns.update((name, rows[k]) for k, name in enumerate(STATE_COLS))
        columns['lossfrac'] = rows[8]
        return pd.DataFrame(columns, index=pd.Index(labels), copy=False)   # one frame, no copies

    def _spill(self):
        """Called before the store is evicted from HBM: keep the rows on the host instead."""
        if self._store is not None:
            if self._X is None:
                self._X = self._frame_from_rows()
            self._store = None

    # ---- drivers --------------------------------------------------------------------------
    def _device_setup(self, ctx):
        ctx.set_forces(**self.forces_kwargs())
        ctx.set_bounce(self._bounce)
        ctx.set_bodies(self._bodies)
        ctx.set_first_index(self._first_index)

    def constant_step_size_driver(self, keep_trajectory=True):
        """Output.py:368-455 on the GPU.  compress=True (the default): the trajectories stay in
        HBM as the rows save() keeps (frac > 0, Output.py:523-524) and ``X`` is assembled from
        them on first access; compress=False: the dense (N*nsteps rows) frame of the reference,
        columns Index,time,x,y,z,vx,vy,vz,frac,lossfrac.  lossfrac starts from 0 (the
        reference's starts from uninitialised memory, Output.py:378)."""
        opt = self.inputs.options
        endtime, step = opt.endtime.value, float(opt.step_size)
        self.nsteps, n_iter = n_output_steps(endtime, step)
        ctx = self.context()
        self._device_setup(ctx)
        self.upload(ctx)
        n = self.npackets
        if keep_trajectory and self.compress:
            self._rows_pass(ctx, [self], step, n_iter)
        elif keep_trajectory:
            res = ctx.integrate_const(step, n_iter, opt.outeredge, nrec=self.nsteps)
            self._raise_on_counters(ctx.counters())
            traj = res['traj']                               # (8, nsteps, N)
            X = pd.DataFrame()
            X['Index'] = np.repeat(np.arange(n, dtype=np.int64), self.nsteps)
            for k, name in enumerate(STATE_COLS):
                X[name] = np.ascontiguousarray(traj[k].T).reshape(n*self.nsteps)
            frac = traj[7].T                                 # (N, nsteps)
            # lossfrac[:, ct] = lossfrac[:, ct-1] + frac[:, ct-1] - frac[:, ct] while the
            # packet was active at ct-1 (Output.py:420-421), evaluated left to right
            lossfrac = np.zeros_like(frac)
            for ct in range(1, self.nsteps):
                act = frac[:, ct-1] > 0
                if not act.any():
                    break
                lossfrac[act, ct] = (lossfrac[act, ct-1] + frac[act, ct-1]) - frac[act, ct]
            X['lossfrac'] = lossfrac.reshape(n*self.nsteps)
            self.X = X
        else:
            ctx.integrate_const(step, n_iter, opt.outeredge)
            self._raise_on_counters(ctx.counters())
            self.X = pd.DataFrame()
        self.totalsource *= self.nsteps                      # Output.py:434
        self._add_units()

    @staticmethod
    def _rows_pass(ctx, outputs, step, n_iter):
        """The compact-rows protocol over the context's resident packets, which are the packets
        of ``outputs`` one Output after the other: the kernel delivers exactly the rows with
        frac > 0, packet-major like the reference's filtered frame, and they stay in HBM; each
        Output gets its slice."""
        lead = outputs[0]
        res = ctx.integrate_const_rows(step, n_iter, lead.inputs.options.outeredge,
                                       narrow=lead._narrow_rows, resident=True)
        ctr = ctx.counters()
        lengths, store = res['lengths'], res['store']
        row0 = packet0 = 0
        for out in outputs:
            out._raise_on_counters(ctr)
            assert ctr.get('unfinished', 0) == 0, 'row passes disagree'
            mine = lengths[packet0:packet0 + out.npackets]
            out._attach_rows(store, row0, mine, packet0)
            row0 += out._nrows
            packet0 += out.npackets
        assert row0 == store.total and packet0 == len(lengths)

    def variable_step_size_driver(self):
        """Output.py:221-366 on the GPU: final snapshot, one row per packet."""
        opt = self.inputs.options
        assert self._bounce is None, 'Not set up'            # Output.py:312-315
        ctx = self.context()
        ctx.set_forces(**self.forces_kwargs())
        ctx.set_bodies(None)
        ctx.upload_soa(np.ascontiguousarray(self.X[STATE_COLS].values.T, dtype=np.float64))
        final, hs = ctx.integrate_var(float(opt.resolution), opt.outeredge)
        self._finish_variable(ctx.counters(), final, hs)

    def _finish_variable(self, ctr, final, hs):
        self._raise_on_counters(ctr)
        assert ctr.get('unfinished', 0) == 0, 'variable-step integration did not finish'
        for k, name in enumerate(STATE_COLS):
            self.X[name] = final[:, k]
        self.X['step_size'] = hs
        self.X['Index'] = self.X.index
        self._add_units()

    @classmethod
    def integrate_batch(cls, outputs, ctx, save=True):
        """Integrate several Outputs of the same inputs (built with ``integrate=False``) in ONE
        device launch: packets are independent, so the chunks of an Input.run (Input.py:243-246)
        only differ in which rows of the result they own.  The kernels' queue then holds all the
        chunks' packets and the long-lived ones of every chunk start first, which is what keeps
        the lanes busy (one reference-sized chunk alone has fewer packets than the chip has
        lanes).  Results are identical to integrating the Outputs one by one."""
        lead = outputs[0]
        opt = lead.inputs.options
        for out in outputs:
            out._ctx = ctx
            out._narrow_rows = bool(save)
        if opt.step_size == 0:
            print('Running variable step size integrator.')
            for out in outputs:
                out.X = out.X0.drop(['longitude', 'latitude', 'local_time'], axis=1, errors='ignore')
                out.X['lossfrac'] = np.zeros(out.npackets)
                assert out._bounce is None, 'Not set up'
            ctx.set_forces(**lead.forces_kwargs())
            ctx.set_bodies(None)
            ctx.upload_soa(np.concatenate([out.x0_soa() for out in outputs], axis=1))
            final, hs = ctx.integrate_var(float(opt.resolution), opt.outeredge)
            ctr, at = ctx.counters(), 0
            for out in outputs:
                out._finish_variable(ctr, final[at:at + out.npackets], hs[at:at + out.npackets])
                at += out.npackets
        else:
            print('Running constant step size integrator.')
            endtime, step = opt.endtime.value, float(opt.step_size)
            nsteps, n_iter = n_output_steps(endtime, step)
            lead._device_setup(ctx)
            if not all(getattr(out, '_resident', False) or out.sampler == 'device'
                       for out in outputs):
                if hasattr(ctx, 'upload_soa_pieces'):
                    ctx.upload_soa_pieces([out.x0_soa() for out in outputs])
                else:
                    ctx.upload_soa(np.concatenate([out.x0_soa() for out in outputs], axis=1))
            cls._rows_pass(ctx, outputs, step, n_iter)
            for out in outputs:
                out.nsteps = nsteps
                out.totalsource *= nsteps                    # Output.py:434
                out._add_units()
        if save:
            for out in outputs:
                out.save()

    def _add_units(self):
        # Output.py:363-366,452-455: aplanet in au, vrplanet in km/s, GM in R^3/s^2
        self.aplanet = Quantity(self.aplanet, 'au')
        self.vrplanet = Quantity(float(self.vrplanet)*self.unit_km, 'km/s')
        self.GM = Quantity(self.GM, 'R3/s2')

    def vrplanet_Rs(self):
        """vrplanet back in R/s, as ModelImage.create_image converts it (ModelImage.py:242-243)."""
        v = self.vrplanet
        return float(v)/self.unit_km if isinstance(v, Quantity) and v.unit == 'km/s' else float(v)

    # ---- persistence (file catalogue instead of PostgreSQL + pickle) ------------------------
    def save(self):
        """Apply the reference's on-disk transformations (compress filter, 32-bit down-cast,
        Output.py:522-543), register in the inputs' catalogue, optionally write an .npz.  Rows
        that came from the compact-rows kernels are already filtered (on the 64-bit frac, like
        the reference: a float32 underflow of frac must not drop a row) and narrowed -- when the
        Output was made to be saved.  One made with save=False keeps 64-bit rows in HBM; saving it
        after all brings them to the host and narrows them there, so that what is catalogued (and
        what ModelImage / LOSResult bin) is float32 / int32 as Output.py:528-543 has it."""
        if self._store is not None and self._X is None and not self._store.narrow:
            self._spill()
        if self._store is None or self._X is not None:
            if self._store is None and self.compress and len(self.X) > 0 and 'frac' in self.X:
                keep = self.X.frac.values > 0
                if not keep.all():
                    self.X = self.X[keep]
            self.X = self._recast(self.X, NARROW)
        narrow = self.__dict__.pop('_x0_narrow', None)
        self.X0 = narrow if narrow is not None else self._recast(self.X0, NARROW)
        self.__dict__.pop('_soa_ready', None)              # (64 bytes a packet: not kept)
        catalogue = getattr(self.inputs, '_catalogue', None)
        if catalogue is not None:
            self.idnum = len(catalogue) + 1
            savepath = getattr(self.inputs, 'savepath', None)
            if savepath:
                os.makedirs(savepath, exist_ok=True)
                self.filename = os.path.join(savepath, f'{self.idnum:010d}.npz')
                submit = getattr(self.inputs, '_write_later', None)
                if submit is not None and self._store is not None:
                    submit(self._write, self.filename)   # D2H + file I/O beside the next launch
                else:
                    self._write(self.filename)
            catalogue.append(self)

    def _write(self, filename):
        data = {f'X0.{c}': self.X0[c].values for c in self.X0}
        store = self._store
        if self._X is None and store is not None and store._r is not None:
            # straight from HBM (own copy stream; possibly on the writer thread): the frame itself
            # is not built for a file.  (If the store is spilled meanwhile, the frame exists.)
            try:
                rows, idx = self._host_rows()
            except Exception:
                if self._X is None:
                    raise
                rows = None
        else:
            rows = None
        if rows is not None:
            data['X.Index'] = idx
            data.update({f'X.{c}': rows[k] for k, c in enumerate(STATE_COLS)})
            data['X.lossfrac'] = rows[8]
        else:
            data.update({f'X.{c}': self.X[c].values for c in self.X})
        np.savez(filename, npackets=self.npackets, totalsource=self.totalsource,
                 nsteps=self.nsteps or 0, aplanet=float(self.aplanet),
                 vrplanet_kms=float(self.vrplanet), compress=self.compress, **data)

    @staticmethod
    def _recast(frame, table):
        """The frame with every column whose dtype is a key of ``table`` converted; built in one
        go (assigning the columns back one by one costs pandas a copy and a cache flush each)."""
        if not len(frame.columns) or not any(frame[c].dtype.type in table for c in frame):
            return frame
        columns = {c: (frame[c].values.astype(table[frame[c].dtype.type])
                       if frame[c].dtype.type in table else frame[c].values) for c in frame}
        return pd.DataFrame(columns, index=frame.index, copy=False)

    @classmethod
    def upcast(cls, frame):
        """restore()'s 32 -> 64 bit conversion (Output.py:555-570)."""
        return cls._recast(frame, WIDE)

    IMAGE_COLS = ('x', 'y', 'z', 'vy', 'frac')

    @classmethod
    def image_columns(cls, source):
        """What create_image needs of a stored Output, without restoring the rest: the five
        sample columns as stored (32-bit after save(); the device widens them exactly like
        restore()'s up-cast, Output.py:555-570), aplanet [au] and vrplanet [km/s].  ``source``: a
        catalogued Output or an .npz path."""
        if isinstance(source, cls):
            frame = source.X
            if len(frame) == 0 or 'x' not in frame:
                return None, float(source.aplanet), float(source.vrplanet)
            columns = [frame[c].values for c in cls.IMAGE_COLS]
            return columns, float(source.aplanet), float(source.vrplanet)
        with np.load(source, allow_pickle=False) as data:
            if 'X.x' not in data.files:
                return None, float(data['aplanet']), float(data['vrplanet_kms'])
            columns = [data['X.' + c] for c in cls.IMAGE_COLS]
            return columns, float(data['aplanet']), float(data['vrplanet_kms'])

    @classmethod
    def restore(cls, source):
        """Return an Output with 64-bit columns from a catalogued Output or an .npz file."""
        if isinstance(source, cls):
            out = source
            out.X0, out.X = cls.upcast(out.X0), cls.upcast(out.X)
            return out
        out = cls.__new__(cls)
        out.filename = source
        with np.load(source, allow_pickle=False) as data:
            out.npackets = int(data['npackets'])
            out.totalsource = float(data['totalsource'])
            out.nsteps = int(data['nsteps'])
            out.aplanet = Quantity(float(data['aplanet']), 'au')
            out.vrplanet = Quantity(float(data['vrplanet_kms']), 'km/s')
            out.compress = bool(data['compress'])
            out.X0 = pd.DataFrame({k[3:]: data[k] for k in data.files if k.startswith('X0.')})
            out.X = pd.DataFrame({k[2:]: data[k] for k in data.files if k.startswith('X.')})
        out.X0, out.X = cls.upcast(out.X0), cls.upcast(out.X)
        return out
