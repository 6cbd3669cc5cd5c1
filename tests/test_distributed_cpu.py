"""N > 1 logic on CPU: two gloo ranks shard the packets by index, each integrates its shard (the
C oracle stands in for the GPU kernel here), and the summed image pair equals the single-rank
result -- counts exactly, weights to fp64 summation order.  Also exercises the ControlPlane
primitives bench.py uses (barrier, MAX/SUM, unique-id broadcast)."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from nexoclom_amd.distributed import ControlPlane, shard_range
    from oracle import np_oracle as O
    from oracle.c_oracle import COracle
    from tests import helpers as H
    cp = ControlPlane()
    assert cp.world == world and cp.rank == rank
    cp.barrier()
    assert cp.reduce(rank + 1.0, 'MAX') == world
    assert cp.reduce(rank + 1.0, 'SUM') == world*(world+1)/2
    payload = bytes(range(128)) if rank == 0 else b''
    assert cp.bcast_bytes(payload, 128) == bytes(range(128))

    n = 3001                                     # not divisible by the world size
    f = H.mercury_forces('Na', 1.3)
    X0 = H.sample_x0(n, 2024, 50000.)            # every rank draws the same global X0
    lo, hi = shard_range(n, rank, world)
    nsteps, n_iter = O.n_output_steps(50000., 30.)
    co = COracle()
    im = H.image_setup(f, 'radiance', dims=(64, 64))
    desc = co.image_desc(im['M'], f.vrplanet, im['apix'], 'radiance', im['g_tables'],
                         im['xedges'], im['zedges'], downcast=True)
    part = co.integrate_const(f, X0[lo:hi], 30., n_iter, 25., img=desc)
    image, counts = cp.allreduce_images_host(part['image'], part['counts'])
    work = cp.reduce(part['work'], 'SUM')
    if rank == 0:
        full = co.integrate_const(f, X0, 30., n_iter, 25., img=desc)
        assert work == full['work']
        assert np.array_equal(counts, full['counts'])
        np.testing.assert_allclose(image, full['image'], rtol=1e-12)
        open(os.path.join(tmpdir, 'ok'), 'w').write('ok')
    cp.barrier()
    cp.close()


def test_two_rank_sharding_and_image_sum(tmp_path):
    port = 29500 + os.getpid() % 400
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok').exists()


def test_shard_ranges_cover_everything():
    from nexoclom_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
