"""A stand-in for nexoclom_amd.hip_api.Context built on the C oracle (TEST INFRASTRUCTURE).

The CPU tests of the multi-GPU path run the PRODUCT's own partition / merge code
(nexoclom_amd.distributed.sharded_image, ModelImage._stream, chunk_plan) on ranks that have no
GPU; the only thing replaced is the device: this class answers the handful of Context methods that
code calls, integrating and binning with oracle/c/oracle.c.  It lives under tests/ and is never
imported by the package.
"""
import numpy as np

from oracle import np_oracle as O
from oracle.c_oracle import COracle


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
