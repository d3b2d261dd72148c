"""Multi-GPU plumbing: one process per GPU, candidates sharded, ONE gather per batch (RCCL over xGMI).

The reference distributes candidates as Ray tasks and streams pickled results back (superdsm/objects.py:275-281,
superdsm/_aux.py:44-48).  Here every rank holds the (small) image, solves its shard of a batch's candidates and
sends fixed-size records (128 B each) plus its bit-packed masks to rank 0 in a single ``gather`` -- direct
peer -> root transfers, no ring.  Candidate solves need no data-path collective.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_indices(costs, world):
    """Deal candidates to ranks: sort by cost (descending), round-robin with alternating direction (snake) so
    that every rank gets a similar total.  Returns a list of index arrays, one per rank."""
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind='stable')
    shards = [[] for _ in range(world)]
    for pos, idx in enumerate(order):
        rnd, k = divmod(pos, world)
        shards[k if rnd % 2 == 0 else world - 1 - k].append(int(idx))
    return [np.asarray(sorted(s), dtype=np.int64) for s in shards]


class RecordGather:
    """Gathers every rank's record block and mask block to rank 0 (padded to the largest rank's size)."""

    def __init__(self, batch_or_tensors, world, rank, group=None):
        if isinstance(batch_or_tensors, tuple):
            self.records, self.masks = batch_or_tensors
        else:
            self.records, self.masks = batch_or_tensors.records_dev, batch_or_tensors.masks_dev
        self.world, self.rank, self.group = world, rank, group
        dev = self.records.device
        sizes = torch.tensor([self.records.numel(), self.masks.numel()], dtype=torch.int64, device=dev)
        all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(all_sizes, sizes, group=group)
        self.sizes = torch.stack(all_sizes).cpu().numpy()
        self.pad = int(self.sizes.sum(axis=1).max())
        self.send = torch.zeros(self.pad, dtype=torch.uint8, device=dev)
        self.recv = [torch.zeros(self.pad, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None

    def run(self):
        nr, nm = self.records.numel(), self.masks.numel()
        self.send[:nr].copy_(self.records.view(torch.uint8).reshape(-1))
        self.send[nr:nr + nm].copy_(self.masks.view(torch.uint8).reshape(-1))
        dist.gather(self.send, self.recv, dst=0, group=self.group)

    def unpack(self):
        """On rank 0: list of (records bytes, masks bytes) numpy arrays per rank."""
        assert self.rank == 0
        out = []
        for r in range(self.world):
            buf = self.recv[r].cpu().numpy()
            nr, nm = (int(v) for v in self.sizes[r])
            out.append((buf[:nr].copy(), buf[nr:nr + nm].copy()))
        return out
