"""LOSResult: modelled radiance along spacecraft lines of sight, accumulated on the GPU.

Drop-in for data_simulation/LOSResult.py:19-105,202-276 and compute_iteration.py:90-240 of the
reference on the path from catalogued Outputs to ``radiance`` / ``npackets`` per spectrum.  The
reference takes a ``MESSENGERdata`` object (external package, unavailable); anything with the same
duck type works here: ``.data`` (DataFrame with x, y, z, xbore, ybore, zbore in planet radii, model
frame), ``.species``, ``.query``, ``.set_frame()``, ``len()``.  ``SpacecraftData`` is a minimal one
for synthetic geometries.

Division of labour: everything per SPECTRUM that involves libm trigonometry -- the planet cut-off,
the geometric ladder of pre-selection spheres along the line of sight, the cone-angle threshold --
is set up here with NumPy (``los_geometry``, ``arccos_threshold``), so that the thresholds are the
reference's own doubles; every (sample, spectrum) pair test runs in the HIP kernel
(``nxc_los_accumulate``).  The fitted-result machinery (LOSResultFitted, source maps, masks,
PostgreSQL caching of iterations) is out of scope.
"""
import numpy as np
import pandas as pd

from .ModelImage import ModelResult

POSITION = ('x', 'y', 'z')
BORESIGHT = ('xbore', 'ybore', 'zbore')


class SpacecraftData:
    """Minimal stand-in for MESSENGERuvvs.MESSENGERdata: positions and boresights per spectrum."""

    def __init__(self, x, y, z, xbore, ybore, zbore, species='Na', query='synthetic'):
        columns = dict(zip(POSITION + BORESIGHT, (x, y, z, xbore, ybore, zbore)))
        self.data = pd.DataFrame(columns, dtype=float)
        self.species, self.query, self.frame = species, query, 'Model'

    def set_frame(self, frame):
        self.frame = frame

    def __len__(self):
        return len(self.data)


def arccos_threshold(dphi):
    """Smallest double c with arccos(c) <= dphi: the reference's ``ang <= dphi`` with
    ``ang = np.arccos(cosang)`` (compute_iteration.py:181-185) is then exactly ``cosang >= c`` for
    this NumPy's (monotone) arccos, so the kernel needs no arccos."""
    below, reached = 0.0, 1.0                # arccos(below) > dphi >= arccos(reached)
    assert np.arccos(below) > dphi >= np.arccos(reached)
    while True:
        middle = 0.5*(below + reached)
        if middle in (below, reached):
            return reached
        if np.arccos(middle) <= dphi:
            reached = middle
        else:
            below = middle


def ladder_to(limit, first, ratio):
    """t_0 = first, t_{k+1} = t_k + t_k * ratio, up to and including the first rung >= limit
    (compute_iteration.py:164-167: sample points spaced in proportion to their distance)."""
    rungs = [first]
    while rungs[-1] < limit:
        rungs.append(rungs[-1] + rungs[-1] * ratio)
    return rungs


def los_geometry(data, outeredge, dphi):
    """Per-spectrum set-up of compute_iteration.py:101-115,157-167.  Returns
    (dist_from_plan, ladder lengths, longest ladder): the distance at which a line of sight that
    hits the planet is cut (1e30 when it misses), and the ladder out to where the line leaves the
    sphere r = outeredge -- all spectra share one geometric ladder, they differ in its length."""
    at = data[list(POSITION)].values.astype(float)
    look = data[list(BORESIGHT)].values.astype(float)
    x, y, z = at.T
    xb, yb, zb = look.T
    dist_from_plan = np.sqrt(x**2 + y**2 + z**2)
    with np.errstate(invalid='ignore'):
        off_centre = np.arccos((-x*xb - y*yb - z*zb) / dist_from_plan)
        planet_size = np.arcsin(1. / dist_from_plan)
    dist_from_plan = np.where(off_centre > planet_size, 1e30, dist_from_plan)
    step = np.sin(dphi)
    far = np.empty(len(at))
    for k, (x_sc, bore) in enumerate(zip(at, look)):
        # far root of |x_sc + t bore| = outeredge
        b = 2*np.sum(x_sc*bore)
        c = np.linalg.norm(x_sc)**2 - outeredge**2
        with np.errstate(invalid='ignore'):
            far[k] = (-b + np.sqrt(b**2 - 4*1*c))/2
    # every spectrum climbs the same rungs and stops at its own `far`: one ladder to the largest,
    # a spectrum's length = the rungs below its limit plus the one that reaches it (a line that
    # never meets the sphere has a NaN root and, like the reference's loop, just the first rung)
    reach = np.nanmax(far) if np.isfinite(far).any() else step
    ladder = np.array(ladder_to(reach, step, step))
    lengths = np.where(np.isnan(far), 1, np.searchsorted(ladder, far, side='left') + 1)
    lengths = np.minimum(lengths, len(ladder)).astype(np.int64)
    return dist_from_plan, lengths, ladder[:lengths.max()]


class LOSResult(ModelResult):
    def __init__(self, scdata, inputs, params=None, dphi=np.radians(1.), *, device=0,
                 context=None, **kwargs):
        scdata.set_frame('Model')
        super().__init__(inputs, {'quantity': 'radiance'} if params is None else params)
        if self.quantity != 'radiance':
            assert False, 'Other quantities not set up.'      # compute_iteration.py:213
        self.type = 'LineOfSight'
        self.species, self.query = scdata.species, scdata.query
        self.dphi = float(dphi)
        self._oedge = np.min([self.inputs.options.outeredge*2, 100])
        self.fitted = self.inputs.options.fitted
        rows = scdata.data.index
        self.radiance = pd.Series(np.zeros(len(rows)), index=rows)
        self.npackets_los = pd.Series(np.zeros(len(rows), dtype=np.int64), index=rows)
        self.included = None
        self.label = kwargs.get('label', 'LOSResult')
        self._ctx, self._device = context, device
        self.iterations = []
        self._geometry = None

    def context(self):
        if self._ctx is None:
            # the device the catalogued runs were made on, when there is one: their rows are still
            # in its HBM (and a new handle costs 0.1 s); else a fresh one
            shared = [getattr(run, '_ctx', None) for run in getattr(self.inputs, '_catalogue', ())]
            shared = [ctx for ctx in shared if ctx is not None and getattr(ctx, '_h', True)]
            if shared:
                self._ctx = shared[-1]
            else:
                from . import hip_api
                self._ctx = hip_api.Context(self._device)
        return self._ctx

    def compute_iteration(self, output, scdata, used_cap=0):
        """One catalogued Output against all spectra (compute_iteration.py:90-240)."""
        from .Output import Output
        if not isinstance(output, Output):       # a file; a catalogued Output is used as stored:
            output = Output.restore(output)      # the binding widens just the columns it sends
        spectra = scdata.data
        if self._geometry is None or self._geometry[0] is not spectra:
            # the same for every Output of a run (compute_iteration.py recomputes it per file)
            self._geometry = (spectra, los_geometry(spectra, self.inputs.options.outeredge,
                                                    self.dphi))
        cut, lengths, ladder = self._geometry[1]
        sc = np.vstack([spectra[list(POSITION + BORESIGHT)].values.T.astype(float), cut,
                        lengths.astype(float)])
        setup = (self.dphi, np.sin(self.dphi), np.sin(self.dphi*2), arccos_threshold(self.dphi),
                 float(output.vrplanet)/self.unit_km, self.unit_km*1e5,
                 self.g_tables(float(output.aplanet)), ladder, sc)
        view = output.resident_rows(self.context())
        if view is not None:
            # the Output's rows are still in HBM: no host round trip
            store, first, count, packet0 = view
            res = self.context().los_accumulate(*setup, n_index=int(output.npackets),
                                                used_cap=used_cap,
                                                rows=(store, first, count, packet0))
        else:
            samples = output.X
            if 'Index' in samples.columns:
                index = samples['Index'].values
            else:
                index = np.arange(len(samples))
            n_index = int(len(output.X0)) if len(output.X0) else int(index.max()) + 1
            res = self.context().los_accumulate(
                *setup, *(samples[c].values for c in ('x', 'y', 'z', 'vy', 'frac')),
                index=index, n_index=n_index, used_cap=used_cap)
        assert self.context().counters()['nonfinite'] == 0, 'Non-finite weights'
        res['totalsource'] = output.totalsource
        for key in ('radiance', 'npackets'):
            res[key] = pd.Series(res[key], index=spectra.index)
        return res

    def simulate_data_from_inputs(self, scdata, distribute=None, *, cp=None, reduce='rccl'):
        """LOSResult.py:202-276: sum the iterations of every catalogued Output, then scale to kR
        for a source rate of 1e23 atoms/s.  ``cp``: the control plane of a shared run
        (``Input.run(..., cp=cp)``): this rank's catalogue is its share of the Outputs; the
        per-spectrum radiances and packet counts and the source totals are summed over the ranks
        where the reference sums over the files (LOSResult.py:264-266) -- S + S + 2 doubles, one
        all-reduce.  ``iterations`` (with their `included` flags) stay those of the local Outputs."""
        if distribute in (True, 'delay', 'delayed'):
            assert False, "Don't do this"
        self.outid, self.outputfiles, self.npackets, self.totalsource = self.inputs.search()
        print(f'LOSResult: {len(self.outid)} output files found.')
        shared = cp is not None and cp.world > 1
        if self.npackets == 0 and not shared:
            raise RuntimeError('No packets found for these Inputs.')
        self.iterations = [self.compute_iteration(run, scdata) for run in self.inputs._catalogue]
        for it in self.iterations:
            assert len(it['radiance']) == len(scdata.data)
            self.radiance += it['radiance']
            self.npackets_los += it['npackets']
        if shared:
            from .distributed import allreduce_small, guarded
            S = len(scdata.data)
            with guarded(cp, self.context()):
                both = allreduce_small(np.concatenate([
                    np.asarray(self.radiance, dtype=float), np.asarray(self.npackets_los, dtype=float),
                    [float(self.totalsource), float(self.npackets)]]), cp, self.context(), reduce)
            self.radiance[:] = both[:S]
            self.npackets_los[:] = np.rint(both[S:2*S]).astype(self.npackets_los.dtype)
            self.totalsource, self.npackets = float(both[2*S]), int(round(both[2*S + 1]))
            if self.npackets == 0:
                raise RuntimeError('No packets found for these Inputs.')
        per_second = self.totalsource / self.inputs.options.endtime.value
        self.atoms_per_packet = 1e23 / per_second
        self.radiance *= self.atoms_per_packet/1e3      # kR
