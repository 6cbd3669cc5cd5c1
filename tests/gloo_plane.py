"""torch.distributed (gloo, CPU) implementation of the control-plane interface of
nexoclom_amd.distributed.ControlPlane -- TEST INFRASTRUCTURE: the product's own control plane is
plain TCP and imports no torch; this one lets the world_size-2 CPU tests drive the product's
partition / merge code over gloo as well."""
import os

import numpy as np


class GlooControlPlane:
    def __init__(self, world=None, rank=None):
        import torch
        import torch.distributed as dist
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.local_rank = int(os.environ.get('LOCAL_RANK', str(self.rank)))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
        if not dist.is_initialized():
            dist.init_process_group(backend='gloo', rank=self.rank, world_size=self.world)
        self.dist, self.torch = dist, torch

    def barrier(self):
        self.dist.barrier()

    def allreduce(self, values, op='SUM'):
        t = self.torch.from_numpy(np.array(values, dtype=np.float64))
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return t.numpy()

    def reduce(self, value, op='SUM'):
        return float(self.allreduce([float(value)], op)[0])

    def bcast_bytes(self, payload, n=None):
        size = self.torch.tensor([len(payload) if self.rank == 0 else 0])
        self.dist.broadcast(size, src=0)
        t = (self.torch.tensor(list(payload), dtype=self.torch.uint8) if self.rank == 0
             else self.torch.zeros(int(size[0]), dtype=self.torch.uint8))
        self.dist.broadcast(t, src=0)
        out = bytes(t.tolist())
        assert n is None or len(out) == n
        return out

    def allgather_bytes(self, payload):
        objs = [None]*self.world
        self.dist.all_gather_object(objs, bytes(payload))
        return objs

    def allreduce_images_host(self, image, counts):
        ti = self.torch.from_numpy(np.array(image, dtype=np.float64))
        tc = self.torch.from_numpy(np.asarray(counts).astype(np.int64))
        self.dist.all_reduce(ti, op=self.dist.ReduceOp.SUM)
        self.dist.all_reduce(tc, op=self.dist.ReduceOp.SUM)
        return ti.numpy(), tc.numpy().astype(np.uint64)

    def close(self):
        if self.dist.is_initialized():
            self.dist.destroy_process_group()
