"""N > 1 logic on CPU, world_size 2, through the PRODUCT's own partition and merge code:
nexoclom_amd.distributed.sharded_image -> ModelImage._stream (chunk grid, per-chunk seeds, row
slices) -> merge_shards -> finalize.  The only stand-in is the device (tests/oracle_context.py:
the C oracle answers the Context calls).  For BOTH samplers the 2-rank result must equal the
1-rank result: packet-count image and totals exactly, weights to fp64 summation order.  Runs once
over the product's TCP control plane and once over torch.distributed gloo."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INPUT = os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input')
PARAMS = {'quantity': 'radiance', 'dims': '64,64', 'width': '8,8'}
N, SEED, CHUNK = 1201, 2024, 500          # 3 chunks; the shard boundary 601 falls inside chunk 1


def _plane(kind, world, rank):
    if kind == 'gloo':
        from tests.gloo_plane import GlooControlPlane
        return GlooControlPlane(world, rank)
    from nexoclom_amd.distributed import ControlPlane
    return ControlPlane(world, rank, timeout=120)


def _worker(rank, world, port, tmpdir, kind):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import contextlib
    import io
    from nexoclom_amd import Input
    from nexoclom_amd.distributed import ControlPlane, sharded_image
    from tests.oracle_context import OracleContext
    cp = _plane(kind, world, rank)
    assert cp.world == world and cp.rank == rank
    cp.barrier()
    assert cp.reduce(rank + 1.0, 'MAX') == world
    assert cp.reduce(rank + 1.0, 'SUM') == world*(world+1)/2
    assert cp.reduce(rank + 1.0, 'MIN') == 1.0
    payload = bytes(range(128)) if rank == 0 else b''
    assert cp.bcast_bytes(payload, 128) == bytes(range(128))
    assert cp.allgather_bytes(b'r%d' % rank) == [b'r%d' % r for r in range(world)]

    inputs = Input(INPUT)
    # a rank draws only the rows it owns of a chunk it shares (WindowGenerator): record the sizes
    from nexoclom_amd.source_distribution import WindowGenerator
    windows, draw = [], WindowGenerator.random

    def spy(self, size):
        windows.append(size)
        return draw(self, size)
    WindowGenerator.random = spy
    for sampler in ('numpy', 'device'):
        ctx = OracleContext()
        with contextlib.redirect_stdout(io.StringIO()):
            part = sharded_image(inputs, PARAMS, N, SEED, cp=cp, context=ctx, sampler=sampler,
                                 packs_per_it=CHUNK, reduce='host')
        if sampler == 'numpy':
            # chunk 1 (rows 500..999) is shared: rank 0 draws its 101 rows, rank 1 its 399, five
            # vectors each (sin lat, lon, speed, sin alt, azimuth); whole chunks are not windowed
            assert windows == [101 if rank == 0 else 399]*5, windows
        # this rank integrated only its own rows, addressed by their GLOBAL index
        want = [(500, 0), (101, 500)] if rank == 0 else [(399, 601), (201, 1000)]
        assert ctx.calls == want, (sampler, rank, ctx.calls)
        if rank == 0:
            one = OracleContext()
            with contextlib.redirect_stdout(io.StringIO()):
                whole = sharded_image(inputs, PARAMS, N, SEED, cp=ControlPlane(1, 0), context=one,
                                      sampler=sampler, packs_per_it=CHUNK, reduce='host')
            assert one.calls == [(500, 0), (500, 500), (201, 1000)]
            assert whole.npackets == part.npackets == N
            assert whole.totalsource == part.totalsource == N*1668
            assert whole.atoms_per_packet == part.atoms_per_packet
            assert whole.packet_image.sum() > 1000
            assert np.array_equal(whole.packet_image, part.packet_image), sampler
            np.testing.assert_allclose(part.image, whole.image, rtol=1e-12, atol=0)
            np.save(os.path.join(tmpdir, f'{sampler}.npy'), part.packet_image)
    # an unseeded device-sampled run is still one run: every rank integrates under rank 0's fresh
    # key (recorded on the image), and two such runs do not repeat each other
    keys = []
    for _ in range(2):
        ctx = OracleContext()
        with contextlib.redirect_stdout(io.StringIO()):
            img = sharded_image(inputs, PARAMS, 300, None, cp=cp, context=ctx, sampler='device',
                                packs_per_it=CHUNK, reduce='host')
        assert cp.allgather_bytes(str(img.seed).encode()) == [str(img.seed).encode()]*world
        keys.append(img.seed)
    assert keys[0] != keys[1] and all(0 <= k < 2**64 for k in keys)
    cp.barrier()
    cp.close()
    if rank == 0:
        # the two samplers draw different packets: the images must not be the same array
        a, b = (np.load(os.path.join(tmpdir, f'{s}.npy')) for s in ('numpy', 'device'))
        assert not np.array_equal(a, b)
        open(os.path.join(tmpdir, 'ok'), 'w').write('ok')


def _run(kind, tmp_path, world=2):
    port = 29500 + os.getpid() % 400 + (17 if kind == 'gloo' else 0)
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), kind))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    for p in procs:
        if p.is_alive():
            p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert (tmp_path / 'ok').exists()


def _worker8(rank, world, port, tmpdir, broken):
    """World of eight through sharded_image: every rank integrates its index range, everybody ends
    with the global image.  broken >= 0: that rank's device fails inside its shard."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import contextlib
    import io
    import json
    from nexoclom_amd import Input
    from nexoclom_amd.distributed import ControlPlane, shard_range, sharded_image
    from tests.oracle_context import OracleContext

    class Ctx(OracleContext):
        def integrate_const(self, *a, **k):
            if rank == broken:
                raise RuntimeError('device lost inside the shard')
            return super().integrate_const(*a, **k)
    cp = ControlPlane(world, rank, timeout=60)
    inputs = Input(INPUT)
    ctx = Ctx(threads=1)
    verdict = {'rank': rank}
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            part = sharded_image(inputs, PARAMS, N, SEED, cp=cp, context=ctx, sampler='device',
                                 packs_per_it=CHUNK, reduce='host')
        lo, hi = shard_range(N, rank, world)
        assert sum(n for n, _ in ctx.calls) == hi - lo and ctx.calls[0][1] == lo
        verdict.update(ok=True, counts=float(part.packet_image.sum()), npackets=part.npackets)
        if rank == 0:
            one = OracleContext()
            with contextlib.redirect_stdout(io.StringIO()):
                whole = sharded_image(inputs, PARAMS, N, SEED, cp=ControlPlane(1, 0), context=one,
                                      sampler='device', packs_per_it=CHUNK, reduce='host')
            assert np.array_equal(whole.packet_image, part.packet_image)
            np.testing.assert_allclose(part.image, whole.image, rtol=1e-12, atol=0)
            assert whole.totalsource == part.totalsource
    except Exception as exc:                        # noqa: BLE001 -- the verdict is the test
        verdict.update(ok=False, error=f'{type(exc).__name__}: {exc}', peer=cp.failure)
    cp.close()
    json.dump(verdict, open(os.path.join(tmpdir, f'v{rank}.json'), 'w'))


def _run8(tmp_path, broken=-1, limit=300):
    import json
    port = 29100 + os.getpid() % 300
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, str(tmp_path), broken))
             for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(limit)
    hung = [r for r, p in enumerate(procs) if p.is_alive()]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert not hung, f'ranks {hung} still waiting after {limit} s'
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return [json.load(open(tmp_path / f'v{r}.json')) for r in range(8)]


def test_eight_ranks_equal_one_rank(tmp_path):
    res = _run8(tmp_path)
    assert all(v['ok'] for v in res), res
    assert len({v['counts'] for v in res}) == 1 and res[0]['counts'] > 1000
    assert all(v['npackets'] == N for v in res)


def test_a_failing_rank_ends_the_sharded_image_on_every_rank(tmp_path):
    """Rank 5's device fails inside its shard: no rank is left waiting in the merge; rank 5
    reports its own error, the others a broken collective -- and the watcher knows who it was."""
    res = _run8(tmp_path, broken=5, limit=120)
    assert not any(v['ok'] for v in res)
    assert 'device lost inside the shard' in res[5]['error']
    assert any(v['peer'] and 'rank 5' in v['peer'] for r, v in enumerate(res) if r != 5)


def _worker_two_stage(rank, world, port, tmpdir, kind):
    """The reference's two-stage flow shared by `world` ranks: Input.run(cp=...) gives every rank
    its Outputs of the single-process plan, produce_image(cp=...) and
    LOSResult.simulate_data_from_inputs(cp=...) sum over the ranks what the reference sums over
    the files.  Rank 0 also runs everything alone and compares."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import contextlib
    import io
    from nexoclom_amd import Input, LOSResult, SpacecraftData
    from nexoclom_amd.distributed import ControlPlane, shard_range
    from tests.oracle_context import OracleContext
    cp = _plane(kind, world, rank)

    def inputs_():
        inputs = Input(INPUT)
        inputs.options.endtime = type(inputs.options.endtime)(6000., 's')
        return inputs
    rng = np.random.default_rng(4)
    th = np.linspace(0, 2*np.pi, 40, endpoint=False)
    pos = np.stack([0.5*np.cos(th)*2, 2*np.sin(th)*0.6 - 0.4, 2*np.sin(th)*0.8], 1)
    look = -pos + 0.6*rng.normal(size=pos.shape)
    look /= np.linalg.norm(look, axis=1)[:, None]
    sc = SpacecraftData(pos[:, 0], pos[:, 1], pos[:, 2], look[:, 0], look[:, 1], look[:, 2])
    npackets, size, passes = 2300, 500, 5                 # 5 Outputs of 500: the last overshoots

    def flow(cp_, ctx, sampler='numpy'):
        inputs = inputs_()
        with contextlib.redirect_stdout(io.StringIO()):
            inputs.run(npackets, packs_per_it=size, seed=77, context=ctx, cp=cp_, sampler=sampler)
            image = inputs.produce_image(PARAMS, context=ctx, cp=cp_, reduce='host')
            los = LOSResult(sc, inputs, dphi=np.radians(3.0), context=ctx)
            los.simulate_data_from_inputs(sc, cp=cp_, reduce='host')
        return inputs, image, los
    ctx = OracleContext()
    inputs, image, los = flow(cp, ctx)
    lo, hi = shard_range(passes, rank, world)
    assert len(inputs._catalogue) == hi - lo                       # only this rank's Outputs
    assert [o.npackets for o in inputs._catalogue] == [size]*(hi - lo)
    # ... integrated in a few launches (whatever was drawn when the device was free)
    assert sum(n for n, _ in ctx.calls) == size*(hi - lo) and all(n % size == 0 for n, _ in ctx.calls)
    assert los.npackets == size*passes and image.totalsource == los.totalsource
    if rank == 0:
        alone, image1, los1 = flow(None, OracleContext())
        assert len(alone._catalogue) == passes
        # Output k is the same Output whoever made it
        for k, out in enumerate(inputs._catalogue):
            same = alone._catalogue[lo + k]
            assert np.array_equal(out.X0.values, same.X0.values)
            assert np.array_equal(out.X.values, same.X.values)
        assert image1.totalsource == image.totalsource == size*passes*201
        assert image1.packet_image.sum() > 1000
        assert np.array_equal(image.packet_image, image1.packet_image)
        np.testing.assert_allclose(image.image, image1.image, rtol=1e-12, atol=0)
        assert image.atoms_per_packet == image1.atoms_per_packet
        assert los1.npackets_los.sum() > 100
        assert np.array_equal(los.npackets_los.values, los1.npackets_los.values)
        np.testing.assert_allclose(los.radiance.values, los1.radiance.values, rtol=1e-12, atol=0)
    # the device sampler is counter-based on the GLOBAL packet index: a rank's Outputs must start at
    # the index they have in the one-process run (Input.run's `drawn` under sharding)
    ctx_d = OracleContext()
    inputs_d, image_d, _ = flow(cp, ctx_d, sampler='device')
    firsts = [o._first_index for o in inputs_d._catalogue]
    assert firsts == [size*(lo + k) for k in range(hi - lo)], firsts
    if rank == 0:
        alone_d, image_d1, _ = flow(None, OracleContext(), sampler='device')
        for k, out in enumerate(inputs_d._catalogue):
            assert np.array_equal(out.X0.values, alone_d._catalogue[lo + k].X0.values)
        assert np.array_equal(image_d.packet_image, image_d1.packet_image)
        assert not np.array_equal(image_d.packet_image, image.packet_image)   # other packets than the host's
        open(os.path.join(tmpdir, 'ok'), 'w').write('ok')
    cp.barrier()
    cp.close()


@pytest.mark.parametrize('kind,world', [('tcp', 2), ('gloo', 2), ('tcp', 3), ('tcp', 8)])   # 8: three ranks without an Output
def test_two_stage_flow_and_los_shared_by_ranks_equal_one_rank(tmp_path, kind, world):
    port = 29300 + os.getpid() % 300 + 7*world + (11 if kind == 'gloo' else 0)
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker_two_stage, args=(r, world, port, str(tmp_path), kind))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    for p in procs:
        if p.is_alive():
            p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert (tmp_path / 'ok').exists()


def test_two_ranks_equal_one_rank_over_the_tcp_control_plane(tmp_path):
    _run('tcp', tmp_path)


def test_two_ranks_equal_one_rank_over_gloo(tmp_path):
    _run('gloo', tmp_path)


def test_shard_ranges_cover_everything():
    from nexoclom_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_chunk_plan_is_independent_of_the_shard_count():
    """The union over ranks of the per-rank pieces is the single-rank chunk loop, every global
    index exactly once, each piece inside one chunk."""
    from nexoclom_amd.distributed import chunk_plan, shard_range
    for total, chunk in ((0, 5), (1, 5), (23, 5), (25, 5), (1201, 500), (10, 100)):
        whole = list(chunk_plan(total, chunk))
        assert [(c0, clen) for _, c0, clen, _, _ in whole] == \
            [(c0, min(chunk, total - c0)) for c0 in range(0, total, chunk)]
        assert all(a == c0 and b == c0 + clen for _, c0, clen, a, b in whole)
        for world in (1, 2, 3, 8):
            seen = np.zeros(total, dtype=int)
            for r in range(world):
                lo, hi = shard_range(total, r, world)
                for k, c0, clen, a, b in chunk_plan(total, chunk, lo, hi):
                    assert c0 == k*chunk and c0 <= a < b <= c0 + clen and lo <= a and b <= hi
                    seen[a:b] += 1
            assert (seen == 1).all()


def test_control_plane_refuses_a_missing_rank_zero(tmp_path):
    from nexoclom_amd.distributed import ControlPlane
    with pytest.raises(TimeoutError):
        ControlPlane(2, 1, timeout=0.3, rendezvous=str(tmp_path / 'nobody.addr'))
