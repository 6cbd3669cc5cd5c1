"""Multi-GPU plumbing: one process per GPU, packets sharded by index, one sum of the image pair.

The reference has no parallelism; its natural data-parallel axis is the
packet index, and the only cross-chunk operation it performs is the per-output-file image sum of
ModelImage.__init__ (data_simulation/ModelImage.py:96-98).  That sum is the one collective here:
``ncclAllReduce`` on the device images, issued from libnexoclom_hip.so (hip_api.Context.
image_allreduce).  ``ControlPlane`` is the CPU-side side channel (torch.distributed gloo):
rendezvous of the RCCL unique id, barriers and scalar reductions -- no packet or pixel data goes
through it, except in ``allreduce_images_host`` which the CPU tests use to exercise the N > 1
logic without GPUs.
"""
import os

import numpy as np


def shard_range(n, rank, world):
    """Contiguous index range [lo, hi) of rank's packets: sizes differ by at most one."""
    base, extra = divmod(int(n), int(world))
    lo = rank*base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ControlPlane:
    """Barrier / scalar reductions / byte broadcast across ranks; trivial when world == 1."""

    def __init__(self, world=None, rank=None, backend='gloo'):
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.local_rank = int(os.environ.get('LOCAL_RANK', str(self.rank)))
        self.dist = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29511')
            if os.environ['MASTER_ADDR'] in ('127.0.0.1', 'localhost'):
                # single node: keep gloo on the loopback device (the hostname may not resolve)
                os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
            if not dist.is_initialized():
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.dist, self.torch = dist, torch

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def reduce(self, value, op='SUM'):
        if not self.dist:
            return value
        t = self.torch.tensor([float(value)], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return float(t[0])

    def bcast_bytes(self, payload, n):
        if not self.dist:
            return payload
        if self.rank == 0:
            t = self.torch.tensor(list(payload), dtype=self.torch.uint8)
        else:
            t = self.torch.zeros(n, dtype=self.torch.uint8)
        self.dist.broadcast(t, src=0)
        return bytes(t.tolist())

    def allreduce_images_host(self, image, counts):
        """Sum an (image fp64, counts uint64) pair over ranks on the HOST (tests / diagnostics;
        the production path is the RCCL all-reduce on the device buffers)."""
        if not self.dist:
            return image, counts
        ti = self.torch.from_numpy(np.ascontiguousarray(image, dtype=np.float64))
        tc = self.torch.from_numpy(np.ascontiguousarray(counts).astype(np.int64))
        self.dist.all_reduce(ti, op=self.dist.ReduceOp.SUM)
        self.dist.all_reduce(tc, op=self.dist.ReduceOp.SUM)
        return ti.numpy(), tc.numpy().astype(np.uint64)

    def init_rccl(self, ctx):
        """Create the RCCL communicator on a hip_api.Context (rank 0's unique id is broadcast
        over the control plane).  Returns True when every rank has a communicator; on False the
        caller must use allreduce_images_host (and say so in what it reports)."""
        from . import hip_api
        ok, uid = 1.0, bytes(hip_api.NXC_UNIQUE_ID_BYTES)
        if self.rank == 0:
            try:
                uid = ctx.comm_unique_id()
            except hip_api.HipError as err:
                print(f'[nexoclom_amd] RCCL unavailable: {err}')
                ok = 0.0
        uid = self.bcast_bytes(uid, hip_api.NXC_UNIQUE_ID_BYTES)
        if self.reduce(ok, 'MIN') < 1.0:
            return False
        try:
            ctx.comm_init(uid, self.rank, self.world)
        except hip_api.HipError as err:
            print(f'[nexoclom_amd] rank {self.rank}: RCCL communicator failed: {err}')
            ok = 0.0
        return self.reduce(ok, 'MIN') >= 1.0

    def close(self):
        if self.dist and self.dist.is_initialized():
            self.dist.destroy_process_group()
            self.dist = None


def sharded_image(inputs, params, npackets, seed, cp=None, device=None, downcast=True,
                  sampler='device', packs_per_it=None):
    """Multi-GPU ModelImage: every rank integrates and bins its contiguous shard of the global
    packet index range, then the image pairs are summed over RCCL (ModelImage.py:96-98 across
    GPUs).  With the counter-based device sampler, packet i is the same packet whatever the
    number of ranks, so the packet-count image is identical for 1, 2, 4 or 8 GPUs.

    Launch one process per GPU (torchrun); returns a ModelImage holding the global image on
    every rank."""
    from . import hip_api
    from .ModelImage import ModelImage
    cp = cp or ControlPlane()
    ndev = hip_api.device_count()
    ctx = hip_api.Context((cp.local_rank if device is None else device) % max(ndev, 1))
    lo, hi = shard_range(int(npackets), cp.rank, cp.world)
    img = ModelImage(inputs, params, npackets=hi - lo, seed=seed, context=ctx, downcast=downcast,
                     sampler=sampler, packs_per_it=packs_per_it, first_index=lo,
                     finalize=False)
    if cp.world > 1:
        if cp.init_rccl(ctx):
            ctx.image_allreduce()
            image, counts = ctx.image_download()
            ctx.comm_destroy()
        else:
            image, counts = cp.allreduce_images_host(*ctx.image_download())
        img.image = image
        img.packet_image = counts.astype(float)
        img.totalsource = cp.reduce(img.totalsource, 'SUM')
        img.npackets = int(cp.reduce(img.npackets, 'SUM'))
    img.finalize()
    return img
