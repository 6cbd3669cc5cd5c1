"""SURVEY 8(e) with real processes on a real GPU: N ranks started the way torch.distributed.run
starts them (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT), each with its own HIP
context, meet over the product's TCP control plane, integrate their shard of the global packet
index range and merge.  The GPU box has ONE device, so the ranks share it and the image pairs are
summed over the control plane (reduce='host'); RCCL itself refuses two ranks on one device
(test_rccl_refuses_two_ranks_on_one_device) and its N > 1 path needs a multi-GPU node.  What this
pins on hardware: rendezvous, shard arithmetic, per-rank sampling (both samplers), merge,
finalize -- the packet-count image of 1, 2 and 3 ranks is the same array."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, 'tools', 'shard_worker.py')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _launch(world, npackets, seed, sampler, flow='streaming'):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   TORCHELASTIC_RUN_ID=f'nxc-test-{port}')
        procs.append(subprocess.Popen([sys.executable, WORKER, str(npackets), str(seed), sampler,
                                       'host', flow], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    lines = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=240)
            assert p.returncode == 0, err[-2000:]
            lines.append(json.loads(out.strip().splitlines()[-1]))
    finally:
        # a rank that failed or timed out must not leave its peers behind: each holds a HIP
        # context on the one GPU and would sit in the control plane until its own timeout
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    return lines


@pytest.mark.parametrize('sampler', ['device', 'numpy'])
def test_ranks_as_processes_give_the_one_rank_image(sampler):
    npackets, seed = 20011, 5
    single, = _launch(1, npackets, seed, sampler)
    assert single['npackets'] == npackets and single['binned'] > 1e5
    for world in (2, 3):
        ranks = _launch(world, npackets, seed, sampler)
        assert sorted(r['rank'] for r in ranks) == list(range(world))
        for r in ranks:                       # every rank ends up holding the global image
            assert r['world'] == world
            assert r['counts_sha1'] == single['counts_sha1']
            assert r['binned'] == single['binned'] and r['totalsource'] == single['totalsource']
            assert r['npackets'] == npackets
            assert abs(r['image_sum'] - single['image_sum']) <= 1e-10*abs(single['image_sum'])


@pytest.mark.parametrize('sampler', ['numpy', 'pcg64'])
def test_two_stage_flow_shared_by_processes_gives_the_one_process_result(sampler):
    """Input.run(cp=...) + produce_image(cp=...) + LOSResult.simulate_data_from_inputs(cp=...) with
    2 and 3 real processes on the GPU: every rank catalogues its share of the plan's Outputs (rows
    resident in its own context) and every rank ends with the image and the line-of-sight
    totals of the whole run -- packet counts per pixel and per spectrum identical to one process."""
    npackets, seed = 14000, 9                      # 5 Outputs of 3000 (the last one overshoots)
    single, = _launch(1, npackets, seed, sampler, 'two-stage')
    assert single['outputs_here'] == 5 and single['npackets'] == 15000
    assert single['binned'] > 1e5 and single['los_pairs'] > 100
    for world in (2, 3):
        ranks = _launch(world, npackets, seed, sampler, 'two-stage')
        assert sum(r['outputs_here'] for r in ranks) == 5
        assert sorted(r['outputs_here'] for r in ranks)[0] >= 1
        for r in ranks:
            assert r['counts_sha1'] == single['counts_sha1']
            assert r['los_counts_sha1'] == single['los_counts_sha1']
            assert r['binned'] == single['binned'] and r['totalsource'] == single['totalsource']
            assert r['npackets'] == single['npackets'] and r['los_pairs'] == single['los_pairs']
            assert abs(r['image_sum'] - single['image_sum']) <= 1e-10*abs(single['image_sum'])
            assert abs(r['los_radiance_sum'] - single['los_radiance_sum']) <= \
                1e-10*abs(single['los_radiance_sum'])
