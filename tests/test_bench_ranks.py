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


def _worker(rank, world, port, tmpdir, broken_rank):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bench
    from nexoclom_amd import hip_api
    from nexoclom_amd.distributed import ControlPlane
    from tests.oracle_context import OracleContext
    args = bench.parse(ARGV)
    cp = ControlPlane(world, rank, timeout=120)
    ctx = OracleContext(cp=cp, seat=f'oracle:{rank}')

    def make_context():
        if rank == broken_rank:
            raise hip_api.HipError(f'rank {rank}: device {rank} does not exist (1 visible)')
        return ctx
    lines = []
    rc = bench.run_rank(args, cp, make_context, emit=lines.append)
    out = {'rc': rc, 'lines': lines, 'log': ctx.log, 'calls': ctx.calls,
           'work': ctx._ctr.get('particle_steps'), 'binned': ctx._ctr.get('samples_binned'),
           'counts_sum': None if ctx._counts is None else float(ctx._counts.sum())}
    json.dump(out, open(os.path.join(tmpdir, f'rank{rank}.json'), 'w'))


def _run(tmp_path, broken_rank=-1, world=2):
    port = 29900 + os.getpid() % 300
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), broken_rank))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    for p in procs:
        if p.is_alive():
            p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return [json.load(open(tmp_path / f'rank{r}.json')) for r in range(world)]


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
