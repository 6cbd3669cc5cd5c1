"""LOSResult: modelled radiance along spacecraft lines of sight, accumulated on the GPU.

Mirrors data_simulation/LOSResult.py:19-105,202-276 and compute_iteration.py:90-240 of the
reference for the path from catalogued Outputs to ``radiance`` / ``npackets`` per spectrum.  The
reference takes a ``MESSENGERdata`` object (external package, unavailable); anything with the same
duck type works here: ``.data`` (DataFrame with x, y, z, xbore, ybore, zbore in planet radii, model
frame), ``.species``, ``.query``, ``.set_frame()``, ``len()``.  ``SpacecraftData`` is a minimal one
for synthetic geometries.  The fitted-result machinery (LOSResultFitted, source maps, masks,
PostgreSQL caching of iterations) is out of scope.
"""
import numpy as np
import pandas as pd

from .ModelImage import ModelResult


class SpacecraftData:
    """Minimal stand-in for MESSENGERuvvs.MESSENGERdata: positions and boresights per spectrum."""

    def __init__(self, x, y, z, xbore, ybore, zbore, species='Na', query='synthetic'):
        self.data = pd.DataFrame({'x': x, 'y': y, 'z': z, 'xbore': xbore, 'ybore': ybore,
                                  'zbore': zbore}, dtype=float)
        self.species = species
        self.query = query
        self.frame = 'Model'

    def set_frame(self, frame):
        self.frame = frame

    def __len__(self):
        return len(self.data)


def arccos_threshold(dphi):
    """Smallest double c with arccos(c) <= dphi: the reference's ``ang <= dphi`` with
    ``ang = np.arccos(cosang)`` (compute_iteration.py:181-185) is then exactly ``cosang >= c`` for
    this NumPy's (monotone) arccos, so the kernel needs no arccos."""
    lo, hi = 0.0, 1.0
    assert np.arccos(lo) > dphi >= np.arccos(hi)
    while True:
        mid = 0.5*(lo + hi)
        if mid == lo or mid == hi:
            break
        if np.arccos(mid) <= dphi:
            hi = mid
        else:
            lo = mid
    return hi


def los_geometry(data, outeredge, dphi):
    """Per-spectrum set-up of compute_iteration.py:101-115,157-167: planet cut-off distance (1e30
    when the line of sight misses the planet) and the ladder t_k = t_{k-1}(1 + sin dphi) out to
    the far side of the outer edge.  Returns (dist_from_plan, ladder lengths, longest ladder)."""
    x, y, z = (data[k].values.astype(float) for k in ('x', 'y', 'z'))
    xb, yb, zb = (data[k].values.astype(float) for k in ('xbore', 'ybore', 'zbore'))
    dist_from_plan = np.sqrt(x**2 + y**2 + z**2)
    with np.errstate(invalid='ignore'):
        ang = np.arccos((-x*xb - y*yb - z*zb) / dist_from_plan)
        asize_plan = np.arcsin(1. / dist_from_plan)
    dist_from_plan = dist_from_plan.copy()
    dist_from_plan[ang > asize_plan] = 1e30
    lengths = np.zeros(len(x), dtype=np.int64)
    longest = np.array([np.sin(dphi)])
    for i in range(len(x)):
        x_sc = np.array([x[i], y[i], z[i]])
        bore = np.array([xb[i], yb[i], zb[i]])
        b = 2*np.sum(x_sc*bore)
        c = np.linalg.norm(x_sc)**2 - outeredge**2
        with np.errstate(invalid='ignore'):
            dd = (-b + np.sqrt(b**2 - 4*1*c))/2
        t = [np.sin(dphi)]
        while t[-1] < dd:
            t.append(t[-1] + t[-1] * np.sin(dphi))
        lengths[i] = len(t)
        if len(t) > len(longest):
            longest = np.array(t)
    return dist_from_plan, lengths, longest


class LOSResult(ModelResult):
    def __init__(self, scdata, inputs, params=None, dphi=np.radians(1.), *, device=0,
                 context=None, **kwargs):
        if params is None:
            params = {'quantity': 'radiance'}
        scdata.set_frame('Model')
        super().__init__(inputs, params)
        if self.quantity != 'radiance':
            assert False, 'Other quantities not set up.'      # compute_iteration.py:213
        self.species = scdata.species
        self.query = scdata.query
        self.type = 'LineOfSight'
        self.dphi = float(dphi)
        self._oedge = np.min([self.inputs.options.outeredge*2, 100])
        self.fitted = self.inputs.options.fitted
        nspec = len(scdata)
        self.radiance = pd.Series(np.zeros(nspec), index=scdata.data.index)
        self.npackets_los = pd.Series(np.zeros(nspec, dtype=np.int64), index=scdata.data.index)
        self.included = None
        self.label = kwargs.get('label', 'LOSResult')
        self._ctx = context
        self._device = device
        self.iterations = []

    def context(self):
        if self._ctx is None:
            from . import hip_api
            self._ctx = hip_api.Context(self._device)
        return self._ctx

    def compute_iteration(self, output, scdata, used_cap=0):
        """One catalogued Output against all spectra (compute_iteration.py:90-240)."""
        from .Output import Output
        output = Output.restore(output)
        packets = output.X
        data = scdata.data
        outeredge = self.inputs.options.outeredge
        dist_from_plan, lengths, ladder = los_geometry(data, outeredge, self.dphi)
        sc = np.stack([data.x.values, data.y.values, data.z.values, data.xbore.values,
                       data.ybore.values, data.zbore.values, dist_from_plan,
                       lengths.astype(float)]).astype(float)
        index = (packets['Index'].values if 'Index' in packets.columns
                 else np.arange(len(packets)))
        n_index = int(len(output.X0)) if len(output.X0) else int(index.max()) + 1
        vr = float(output.vrplanet)/self.unit_km
        res = self.context().los_accumulate(
            self.dphi, np.sin(self.dphi), np.sin(self.dphi*2), arccos_threshold(self.dphi), vr,
            self.unit_km*1e5, self.g_tables(float(output.aplanet)), ladder, sc,
            packets['x'].values, packets['y'].values, packets['z'].values, packets['vy'].values,
            packets['frac'].values, index=index, n_index=n_index, used_cap=used_cap)
        ctr = self.context().counters()
        assert ctr['nonfinite'] == 0, 'Non-finite weights'
        res['totalsource'] = output.totalsource
        res['radiance'] = pd.Series(res['radiance'], index=data.index)
        res['npackets'] = pd.Series(res['npackets'], index=data.index)
        return res

    def simulate_data_from_inputs(self, scdata, distribute=None):
        """LOSResult.py:202-276: sum the iterations of every catalogued Output, then scale to kR
        for a source rate of 1e23 atoms/s."""
        if distribute in (True, 'delay', 'delayed'):
            assert False, "Don't do this"
        self.outid, self.outputfiles, self.npackets, self.totalsource = self.inputs.search()
        print(f'LOSResult: {len(self.outid)} output files found.')
        if self.npackets == 0:
            raise RuntimeError('No packets found for these Inputs.')
        self.iterations = []
        for out in self.inputs._catalogue:
            it = self.compute_iteration(out, scdata)
            assert len(it['radiance']) == len(scdata.data)
            self.iterations.append(it)
            self.radiance += it['radiance']
            self.npackets_los += it['npackets']
        model_rate = self.totalsource / self.inputs.options.endtime.value
        self.atoms_per_packet = 1e23 / model_rate
        self.radiance *= self.atoms_per_packet/1e3      # kR
