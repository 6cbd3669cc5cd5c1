"""A stand-in for nexoclom_amd.hip_api.Context built on the C oracle (TEST INFRASTRUCTURE).

The CPU tests of the multi-GPU path run the PRODUCT's own partition / merge code
(nexoclom_amd.distributed.sharded_image, ModelImage._stream, chunk_plan) on ranks that have no
GPU; the only thing replaced is the device: this class answers the handful of Context methods that
code calls, integrating and binning with oracle/c/oracle.c.  It lives under tests/ and is never
imported by the package.
"""
import weakref

import numpy as np

from oracle import np_oracle as O
from oracle.c_oracle import COracle


class OracleRowStore:
    """Stand-in for hip_api.RowStore: the rows save() keeps, held on the host."""

    def __init__(self, ctx, rows, index, narrow):
        self.ctx, self._r, self.narrow = ctx, object(), narrow
        self._rows, self._index = rows, index
        self.total = rows.shape[1]
        self.owners = weakref.WeakSet()

    @property
    def nbytes(self):
        return self._rows.nbytes + self._index.nbytes

    def download(self, first=0, count=None, index=True):
        count = self.total - first if count is None else int(count)
        return (self._rows[:, first:first + count].copy(),
                self._index[first:first + count].copy() if index else None)

    def free(self):
        self._r = None

    def spill(self):
        for owner in list(self.owners):
            owner._spill()
        self.free()


class OracleContext:
    def __init__(self, threads=2, cp=None, seat='oracle:0'):
        """cp + seat: only for code that also drives the communicator (bench.run_rank): the
        collectives of the stand-in go over that control plane, ``seat`` plays the PCI bus id."""
        self.cp, self.seat = cp, seat
        self.log = []                      # communicator / collective calls, in order
        self._last_ms = 0.0
        self.co = COracle()
        self.threads = threads
        self.n_packets = 0
        self.image_shape = None
        self._forces = None
        self._desc = None
        self._soa = None
        self._image = None
        self._counts = None
        self._ctr = {}
        self.calls = []                    # (n_packets, first_index) of every integrate call
        self._first = 0
        self.abort_requested = False

    # -- set-up -----------------------------------------------------------------------------
    def set_forces(self, GM, vrplanet, gravity=True, radpres=True, lifetime=0.0, photo=None,
                   v_tab=None, a_tab=None):
        kw = {}
        if radpres:
            kw = dict(v_tab=np.asarray(v_tab, float), a_tab=np.asarray(a_tab, float))
        self._forces = O.Forces(GM=GM, vrplanet=vrplanet, gravity=gravity, radpres=radpres,
                                lifetime=lifetime, photo=photo, **kw)

    def set_image(self, M, vrplanet, apix_cm2, quantity, xedges, zedges, g_tables=(),
                  downcast_f32=False):
        self._desc = self.co.image_desc(M, vrplanet, apix_cm2, quantity, list(g_tables), xedges,
                                        zedges, downcast=downcast_f32)
        self.image_shape = (len(xedges) - 1, len(zedges) - 1)
        self.image_clear()

    def set_bounce(self, cfg):
        assert cfg is None, 'the stand-in only covers perfect sticking'

    def set_bodies(self, cfg):
        assert cfg is None, 'the stand-in only covers the single-body model'

    def set_first_index(self, first_index):
        self._first = int(first_index)

    # -- packets ----------------------------------------------------------------------------
    def upload_soa(self, soa):
        soa = np.ascontiguousarray(soa, dtype=np.float64)
        assert soa.ndim == 2 and soa.shape[0] == 8
        self._soa = soa
        self.n_packets = soa.shape[1]

    def sample_packets(self, n, seed, first_index=0, download=False, **src):
        X = O.sample_x0_philox(int(n), int(seed), int(first_index), **src)
        self.upload_soa(X.T)
        return self._soa.copy() if download else None

    # -- image ------------------------------------------------------------------------------
    def image_clear(self):
        self._image = np.zeros(self.image_shape)
        self._counts = np.zeros(self.image_shape, dtype=np.uint64)

    def image_download(self):
        return self._image.copy(), self._counts.copy()

    def counters(self):
        return dict(self._ctr)

    # -- what bench.run_rank needs beyond sharded_image ----------------------------------------
    def device_name(self):
        return 'C oracle (test stand-in for a device)'

    def bus_id(self):
        return self.seat

    def synchronize(self):
        pass

    def close(self):
        self.log.append('close')

    def last_kernel_ms(self):
        return self._last_ms

    def integrate_const_async(self, step, n_iter, outeredge, image=True):
        self.integrate_const(step, n_iter, outeredge, image=image)

    def integrate_const_streamed(self, soa, step, n_iter, outeredge, image=True, pieces=16):
        self.upload_soa(soa)
        self.integrate_const(step, n_iter, outeredge, image=image)

    def comm_unique_id(self):
        self.log.append('comm_unique_id')
        return bytes(range(128))

    def comm_init(self, unique_id, rank, nranks):
        assert unique_id == bytes(range(128)) and (rank, nranks) == (self.cp.rank, self.cp.world)
        self.log.append('comm_init')

    def comm_destroy(self):
        self.log.append('comm_destroy')

    def comm_abort(self):
        self.log.append('comm_abort')

    def comm_request_abort(self):          # called from the control plane's watcher thread
        self.abort_requested = True

    def comm_set_timeout(self, seconds):
        self.log.append('comm_set_timeout')

    def allreduce(self, values):
        self.log.append('allreduce')
        return self.cp.allreduce(np.asarray(values, dtype=np.float64))

    def image_allreduce(self):
        self.log.append('image_allreduce')
        self._image, self._counts = self.cp.allreduce_images_host(self._image, self._counts)

    def allreduce_max(self, value):
        self.log.append('allreduce_max')
        return self.cp.reduce(value, 'MAX')

    def allreduce_sum(self, value):
        self.log.append('allreduce_sum')
        return self.cp.reduce(value, 'SUM')

    def barrier(self):
        self.log.append('barrier')
        self.cp.barrier()

    def integrate_const(self, step, n_iter, outeredge, image=False, nrec=0, want_final=False,
                        want_steps=False):
        assert nrec == 0
        import time
        t0 = time.perf_counter()
        X0 = np.ascontiguousarray(self._soa.T)
        res = self.co.integrate_const(self._forces, X0, step, n_iter, outeredge,
                                      img=self._desc if image else None, threads=self.threads)
        if image:
            self._image += res['image']
            self._counts += res['counts']
        self._ctr = dict(particle_steps=res['work'], samples=0,
                         samples_binned=int(res['counts'].sum()) if image else 0, nonfinite=0,
                         bad_step=0, neg_frac=0, unfinished=0)
        self._last_ms = (time.perf_counter() - t0)*1e3
        self.calls.append((self.n_packets, self._first))
        return dict(traj=None, final=res['final'], steps=res['steps'])

    # -- the two-stage flow: Input.run -> catalogue -> produce_image / LOSResult -----------------
    _h = True                                # "alive", for ModelImage.context() / LOSResult.context()

    def mem_info(self):
        return 1 << 40, 1 << 40

    def make_room(self, need):
        pass

    def integrate_const_rows(self, step, n_iter, outeredge, narrow=False, resident=False):
        """The frac > 0 records of the dense trajectory, packet-major, with lossfrac in the
        reference's association (Output.py:420-421) -- what nxc_integrate_const_rows delivers."""
        X0 = np.ascontiguousarray(self._soa.T)
        n, nrec = self.n_packets, n_iter + 1
        res = self.co.integrate_const(self._forces, X0, step, n_iter, outeredge, nrec=nrec,
                                      threads=self.threads)
        traj = res['traj']                                   # (8, nrec, n)
        frac = traj[7].T                                     # (n, nrec)
        lossfrac = np.zeros_like(frac)
        for ct in range(1, nrec):
            act = frac[:, ct - 1] > 0
            if not act.any():
                break
            lossfrac[act, ct] = (lossfrac[act, ct - 1] + frac[act, ct - 1]) - frac[act, ct]
        live = frac > 0
        rows = np.stack([traj[c].T[live] for c in range(8)] + [lossfrac[live]])
        index = np.repeat(np.arange(n), live.sum(axis=1))
        if narrow:
            rows, index = rows.astype(np.float32), index.astype(np.int32)
        self._ctr = dict(particle_steps=res['work'], samples=0, samples_binned=0, nonfinite=0,
                         bad_step=0, neg_frac=0, unfinished=0)
        self.calls.append((self.n_packets, self._first))
        lengths = live.sum(axis=1).astype(np.int64)
        if resident:
            return dict(lengths=lengths, store=OracleRowStore(self, rows, index, narrow))
        return dict(lengths=lengths, rows=rows)

    def image_accumulate(self, x, y, z, vy, frac):
        image, counts = self.co.image(self._desc, *(np.asarray(c, dtype=np.float64)
                                                    for c in (x, y, z, vy, frac)))
        self._image += image
        self._counts += counts
        self._ctr = dict(particle_steps=0, samples=len(x), samples_binned=int(counts.sum()),
                         nonfinite=0, bad_step=0, neg_frac=0, unfinished=0)

    def image_accumulate_rows(self, store, first=0, count=None):
        rows, _ = store.download(first, count, index=False)
        self.image_accumulate(rows[1], rows[2], rows[3], rows[5], rows[7])

    def los_accumulate(self, dphi, sin_dphi, sin_2dphi, cos_threshold, vrplanet, unit_cm, g_tables,
                       ladder, sc, x=None, y=None, z=None, vy=None, frac=None, index=None,
                       n_index=0, used_cap=0, rows=None):
        if rows is not None:
            store, first, count, shift = rows
            r, idx = store.download(first, count)
            x, y, z, vy, frac, index = r[1], r[2], r[3], r[5], r[7], idx.astype(np.int64) - shift
        smp = dict(x=np.asarray(x, float), y=np.asarray(y, float), z=np.asarray(z, float),
                   vy=np.asarray(vy, float), frac=np.asarray(frac, float))
        if index is not None:
            smp['Index'] = np.asarray(index)
        scd = dict(zip(('x', 'y', 'z', 'xbore', 'ybore', 'zbore'), np.asarray(sc)[:6]))
        radiance, npackets, included, used = O.los_iteration(
            smp, scd, dphi, self.los_outeredge, vrplanet, list(g_tables), unit_cm,
            n_index=n_index or None)
        self._ctr = dict(particle_steps=0, samples=0, samples_binned=int(npackets.sum()),
                         nonfinite=0, bad_step=0, neg_frac=0, unfinished=0)
        return dict(radiance=radiance, npackets=npackets, included=included, used=None, n_used=0)

    los_outeredge = 25.0       # (the bench inputfile's; the restatement derives the cut-offs from it)
