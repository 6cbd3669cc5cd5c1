"""bench.py's own rank logic (bench.run_rank: shard = chunk `rank`, communicator bring-up, the
image all-reduce inside every timed pass, RCCL barrier / max / sum, rank 0's line, the failure
line) on TWO CPU processes over the product's TCP control plane, with tests/oracle_context.py as
the device -- so that the first 8-GPU run meets no line of the N > 1 path that has never executed.
What the stand-in replaces is the device and the transport of its collectives; which calls are
made, in which order, and what is computed from their results is bench.py's own code."""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGV = ['--packets', '1500', '--steps', '2', '--warmup', '1', '--dims', '64', '--no-extras',
        '--no-cpu-baseline']


def _worker(rank, world, port, tmpdir, broken_rank, fault):
    """fault: None | 'no-device' (make_context raises on broken_rank) | 'raise' (its second pass
    raises) | 'die' (its second pass ends the process without a word)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bench
    from nexoclom_amd import hip_api
    from nexoclom_amd.distributed import ControlPlane, pick_device
    from tests.oracle_context import OracleContext
    args = bench.parse(ARGV)
    cp = ControlPlane(world, rank, timeout=60)
    # the device pick of a node with one GPU per rank: LOCAL_RANK of `world` visible devices
    real_count, hip_api.device_count = hip_api.device_count, lambda: world
    assert pick_device(cp) == rank
    hip_api.device_count = real_count

    class Ctx(OracleContext):
        def integrate_const_async(self, *a, **k):
            if rank == broken_rank and fault in ('raise', 'die') and len(self.calls) == 1:
                if fault == 'die':
                    os._exit(3)
                raise RuntimeError('device lost in the middle of pass 2')
            return super().integrate_const_async(*a, **k)
    ctx = Ctx(cp=cp, seat=f'0000:{rank:02x}:00.0', threads=1)

    def make_context():
        if rank == broken_rank and fault == 'no-device':
            raise hip_api.HipError(f'rank {rank}: device {rank} does not exist (1 visible)')
        return ctx
    lines = []
    rc = bench.run_rank(args, cp, make_context, emit=lines.append)
    out = {'rc': rc, 'lines': lines, 'log': ctx.log, 'calls': ctx.calls,
           'work': ctx._ctr.get('particle_steps'), 'binned': ctx._ctr.get('samples_binned'),
           'counts_sum': None if ctx._counts is None else float(ctx._counts.sum())}
    json.dump(out, open(os.path.join(tmpdir, f'rank{rank}.json'), 'w'))


def _run(tmp_path, broken_rank=-1, world=2, fault=None, limit=600):
    import time
    port = 29900 + os.getpid() % 300
    ctx = mp.get_context('spawn')
    if broken_rank >= 0 and fault is None:
        fault = 'no-device'
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), broken_rank, fault))
             for r in range(world)]
    t0 = time.monotonic()
    for p in procs:
        p.start()
    for p in procs:
        p.join(max(1.0, limit - (time.monotonic() - t0)))
    hung = [r for r, p in enumerate(procs) if p.is_alive()]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert not hung, f'ranks {hung} were still waiting after {limit} s'
    codes = [p.exitcode for p in procs]
    if fault == 'die':
        assert codes[broken_rank] == 3 and all(c == 0 for r, c in enumerate(codes) if r != broken_rank), codes
    else:
        assert all(c == 0 for c in codes), codes
    return [json.load(open(tmp_path / f'rank{r}.json')) if (tmp_path / f'rank{r}.json').exists()
            else None for r in range(world)]


def test_two_ranks_run_the_whole_bench_logic(tmp_path):
    r0, r1 = _run(tmp_path)
    assert r0['rc'] == r1['rc'] == 0
    assert len(r0['lines']) == 1 and r1['lines'] == []          # rank 0 prints the one line
    line = json.loads(r0['lines'][0])
    # every rank integrated its own chunk of the global grid, addressed by its global index
    passes = 1 + 2 + 2                                           # warmup + timed + incl-H2D
    assert r0['calls'] == [[1500, 0]]*passes and r1['calls'] == [[1500, 1500]]*passes
    assert r0['work'] != r1['work']                              # different packets (seed + rank)
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['warmup'] == 1
    assert line['scaling'] == 'weak' and line['dtype'] == 'f64' and line['vs_baseline'] is None
    assert line['particle_steps_per_pass'] == r0['work'] + r1['work']       # RCCL-sum branch
    np.testing.assert_allclose(line['value'],
                               line['particle_steps_per_pass']/(line['ms_per_step']*1e-3))
    assert line['config']['image_reduce'] == 'rccl-allreduce'
    assert line['config']['parallelism'] == 'packet-shard x2'
    assert line['config']['control_plane'] == 'tcp+rccl'
    assert line['value_incl_h2d'] > 0 and line['ms_per_step_incl_h2d'] > 0
    assert line['h2d_pass'].startswith('pipelined')
    assert line['roofline']['contract_bound'] == 'hbm' and 0 < line['roofline']['frac'] < 1
    assert line['cpu_baseline'] is None and 'with_comm' not in line
    for r in (r0, r1):
        log = r['log']
        assert log[:2] == ['comm_unique_id', 'comm_init'] if r is r0 else log[:1] == ['comm_init']
        assert log.count('comm_init') == 1 and log.count('comm_destroy') == 1
        assert log.count('image_allreduce') == passes            # inside every pass
        assert log.count('barrier') == 2 + 2*2                   # timed bracket + two incl-H2D
        assert log.count('allreduce_max') == 2 and log.count('allreduce_sum') == 2
        assert log[-2:] == ['comm_destroy', 'close']
        # after the last pass both ranks hold the global (summed) packet-count image
        assert r['counts_sum'] == r0['binned'] + r1['binned']


def test_a_rank_without_a_device_stops_the_whole_job_with_a_failure_line(tmp_path):
    r0, r1 = _run(tmp_path, broken_rank=1)
    assert r0['rc'] == r1['rc'] == 1
    line = json.loads(r0['lines'][0])
    assert line['value'] is None and line['ms_per_step'] is None and line['n_gpus'] == 2
    assert 'device 1 does not exist' in line['error']
    assert r1['lines'] == [] and 'comm_init' not in r0['log'] and r0['calls'] == []


def test_with_comm_on_one_rank_takes_the_collective_branches(tmp_path, capsys):
    """--with-comm: a world of one through every N > 1 branch (what `bench.py --with-comm` runs
    on the one-GPU box with the real RCCL)."""
    sys.path.insert(0, ROOT)
    import bench
    from nexoclom_amd.distributed import ControlPlane
    from tests.oracle_context import OracleContext
    args = bench.parse(ARGV + ['--with-comm'])
    cp = ControlPlane(1, 0)
    ctx = OracleContext(cp=cp)
    lines = []
    assert bench.run_rank(args, cp, lambda: ctx, emit=lines.append) == 0
    line = json.loads(lines[0])
    assert line['n_gpus'] == 1 and line['config']['image_reduce'] == 'rccl-allreduce'
    assert set(line['with_comm']) >= {'ms_per_step_with_collectives', 'ms_per_step_plain',
                                      'delta_ms_per_step'}
    # warmup + timed + incl-H2D passes reduce the image; the plain comparison passes do not
    assert ctx.log.count('image_allreduce') == 1 + 2 + 2 and len(ctx.calls) == 1 + 2 + 2 + 2
    assert ctx.log.count('comm_init') == ctx.log.count('comm_destroy') == 1


def test_eight_ranks_run_the_whole_bench_logic(tmp_path):
    """The driver's scaling run is a world of EIGHT: same code, eight seats, eight shards."""
    res = _run(tmp_path, world=8)
    assert [r['rc'] for r in res] == [0]*8
    assert len(res[0]['lines']) == 1 and all(r['lines'] == [] for r in res[1:])
    line = json.loads(res[0]['lines'][0])
    passes = 1 + 2 + 2
    for rank, r in enumerate(res):
        assert r['calls'] == [[1500, 1500*rank]]*passes           # shard = chunk `rank`
        assert r['log'].count('image_allreduce') == passes
        assert r['log'][-2:] == ['comm_destroy', 'close']
        assert r['counts_sum'] == sum(x['binned'] for x in res)   # everybody holds the global image
    assert len({r['work'] for r in res}) == 8                     # eight different packet sets
    assert line['n_gpus'] == 8 and line['config']['parallelism'] == 'packet-shard x8'
    assert line['particle_steps_per_pass'] == sum(r['work'] for r in res)
    np.testing.assert_allclose(line['value'],
                               line['particle_steps_per_pass']/(line['ms_per_step']*1e-3))


def test_a_rank_that_raises_mid_pass_ends_all_eight_ranks_with_a_failure_line(tmp_path):
    """Rank 5 raises inside its second pass while the others wait in that pass's image
    all-reduce: the failure channel of the control plane ends every wait, rank 0 prints the
    `"value": null` line naming rank 5's error, every rank returns 1 -- in seconds, not at a
    job limit."""
    import time
    t0 = time.monotonic()
    res = _run(tmp_path, broken_rank=5, world=8, fault='raise', limit=120)
    assert time.monotonic() - t0 < 120
    assert [r['rc'] for r in res] == [1]*8
    line = json.loads(res[0]['lines'][0])
    assert line['value'] is None and line['n_gpus'] == 8
    assert 'rank 5' in line['error'] and 'device lost in the middle of pass 2' in line['error']
    assert all(r['lines'] == [] for r in res[1:])
    assert 'comm_abort' in res[5]['log']


def test_a_rank_that_dies_mid_pass_ends_all_eight_ranks_with_a_failure_line(tmp_path):
    """The same when rank 5's process simply ends (an OOM kill): no goodbye, the closed socket
    is the signal."""
    res = _run(tmp_path, broken_rank=5, world=8, fault='die', limit=120)
    assert res[5] is None
    assert [r['rc'] for r in res if r is not None] == [1]*7
    line = json.loads(res[0]['lines'][0])
    assert line['value'] is None and 'rank 5' in line['error']
