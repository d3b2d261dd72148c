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

    def __init__(self, batch_or_tensors, world, rank, group=None, sizes=None):
        """sizes: (world, 2) array of every rank's (record bytes, mask bytes) when the caller knows them (replicated plans);
        otherwise they are exchanged once, here -- never per batch."""
        if isinstance(batch_or_tensors, tuple):
            self.records, self.masks = batch_or_tensors
        else:
            self.records, self.masks = batch_or_tensors.records_dev, batch_or_tensors.masks_dev
        self.world, self.rank, self.group = world, rank, group
        dev = self.records.device
        if sizes is not None:
            self.sizes = np.asarray(sizes, np.int64).reshape(world, 2)
            self.pad = int(self.sizes.sum(axis=1).max())
            self.send = torch.zeros(self.pad, dtype=torch.uint8, device=dev)
            self.recv = [torch.zeros(self.pad, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
            return
        sizes = torch.tensor([self.records.numel(), self.masks.numel()], dtype=torch.int64, device=dev)
        all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
        if dist.get_backend(group) == 'gloo' and sizes.is_cuda:
            cs = sizes.cpu()
            all_cpu = [torch.zeros_like(cs) for _ in range(world)]
            dist.all_gather(all_cpu, cs, group=group)
            all_sizes = all_cpu
        else:
            dist.all_gather(all_sizes, sizes, group=group)
        self.sizes = torch.stack(all_sizes).cpu().numpy()
        self.pad = int(self.sizes.sum(axis=1).max())
        self.send = torch.zeros(self.pad, dtype=torch.uint8, device=dev)
        self.recv = [torch.zeros(self.pad, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None

    def run(self):
        nr, nm = self.records.numel(), self.masks.numel()
        self.send[:nr].copy_(self.records.view(torch.uint8).reshape(-1))
        self.send[nr:nr + nm].copy_(self.masks.view(torch.uint8).reshape(-1))
        if dist.get_backend(self.group) == 'gloo' and self.send.is_cuda:
            # rehearsal path only (gloo has no CUDA gather): stage through the host
            send = self.send.cpu()
            recv = [torch.zeros_like(send) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(send, recv, dst=0, group=self.group)
            if self.rank == 0:
                for r in range(self.world):
                    self.recv[r].copy_(recv[r])
            return
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


def _gpu_solve_local(image, footprints, cfg, mask_info):
    """Default local solver of a shard: one engine batch on this rank's GPU.  Returns (records, masks) as uint8 DEVICE tensors
    (nothing is copied to the host here); ``mask_info`` is what every rank expects the shard's mask boxes to be."""
    from . import _capi, engine
    batch = engine.Batch(image, footprints, cfg, latency_mode=True)
    assert np.array_equal(batch.mask_info[:len(footprints)], mask_info), 'plan of the shard disagrees with the replicated layout'
    batch.launch()
    n = len(footprints)
    if n:
        # candidates whose workgroup group was given up (a scheduling event on an oversubscribed GPU, SDSM_CAND_GIVEN_UP) are solved
        # again without groups on the owning rank, before the all-gather -- as objects._solve does for unsharded batches
        rec32 = batch.records_dev.view(torch.int32).reshape(-1, 32)
        again = torch.nonzero(rec32[:n, 16] == _capi.CAND_GIVEN_UP).flatten().cpu().numpy()       # (status: int32 #16 of the 128-byte record)
        if again.size:
            sub = engine.Batch(image, [footprints[i] for i in again], cfg, mode=2)
            sub.launch()
            sub32 = sub.records_dev.view(torch.int32).reshape(-1, 32)
            for j, i in enumerate(again):
                rec32[i] = sub32[j]
                nb = 4 * ((int(batch.mask_info[i, 2]) * int(batch.mask_info[i, 3]) + 31) // 32)
                o, so = int(batch.mask_offset[i]), int(sub.mask_offset[j])
                batch.masks_dev[o:o + nb] = sub.masks_dev[so:so + nb]
            torch.cuda.current_stream().synchronize()                      # (sub's buffers are released on return)
    return batch.records_dev[:n * 128], batch.masks_dev, batch          # (the batch keeps the buffers alive)


def fragments_from_masks(records, mask_info, mask_offset, masks):
    """Foreground fragments from bit-packed region-bbox masks (engine.fragments_from_masks: host code of the C ABI)."""
    from . import engine
    return engine.fragments_from_masks(records, mask_info, mask_offset, masks)


class Sharder:
    """Splits every batch of candidates over the ranks of a process group and all-gathers the results, so that every
    rank can continue the (cheap, deterministic) host-side generation logic in lock step.

    Per batch: each rank solves its cost-balanced share on its own GPU (no data-path collective), then ONE all-gather of a
    padded byte block per rank -- records (128 B / candidate) and the bit-packed masks, packed on the device and gathered
    device to device (RCCL); one download of the gathered block follows.  Nothing else is exchanged: shards, mask boxes and
    every rank's payload size follow from the image statistics that all ranks hold (the plan is host arithmetic)."""

    def __init__(self, group=None, device=None, solve_local=_gpu_solve_local):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device
        self.solve_local = solve_local

    def solve(self, image, footprints, cfg):
        from . import _capi, engine
        n = len(footprints)
        footprints = [sorted(int(a) for a in fp) for fp in footprints]
        area = np.asarray(image.atom_stats).reshape(-1, 6)[:, 0]
        costs = [sum(int(area[a]) for a in fp if 0 < a < len(area)) for fp in footprints]
        shards = shard_indices(costs, self.world)
        # mask boxes of all candidates and with them every rank's payload size: host-only planning, identical on all ranks
        mask_info = engine.plan_mask_boxes(image, footprints, cfg)
        words = (mask_info[:, 2].astype(np.int64) * mask_info[:, 3] + 31) // 32
        nrec = [128 * len(sh) for sh in shards]
        nmask = [max(4, int(4 * words[sh].sum())) for sh in shards]       # (an empty plan still has a 4-byte mask buffer)
        pad = max(a + b for a, b in zip(nrec, nmask))
        mine = shards[self.rank]
        res = self.solve_local(image, [footprints[i] for i in mine], cfg, mask_info[mine])
        rec_t, mask_t = res[0], res[1]
        dev = rec_t.device if self.device is None else torch.device(self.device)
        send = torch.zeros(pad, dtype=torch.uint8, device=dev)
        send[:nrec[self.rank]].copy_(rec_t.reshape(-1)[:nrec[self.rank]])
        send[nrec[self.rank]:nrec[self.rank] + nmask[self.rank]].copy_(mask_t.reshape(-1)[:nmask[self.rank]])
        recv = torch.empty(self.world * pad, dtype=torch.uint8, device=dev)
        if dist.get_backend(self.group) == 'gloo' and send.is_cuda:          # rehearsal on one GPU: gloo moves host memory
            hs, hr = send.cpu(), torch.empty(self.world * pad, dtype=torch.uint8)
            dist.all_gather(list(hr.chunk(self.world)), hs, group=self.group)
            host = hr.numpy()
        else:
            dist.all_gather(list(recv.chunk(self.world)), send, group=self.group)   # the one collective of the batch
            host = recv.cpu().numpy()
        records = np.zeros(n, _capi.RECORD_DTYPE)
        fragments = [None] * n
        for r in range(self.world):
            idx = shards[r]
            if len(idx) == 0:
                continue
            buf = host[r * pad:(r + 1) * pad]
            rec = buf[:nrec[r]].view(_capi.RECORD_DTYPE)
            ofs = np.concatenate([[0], np.cumsum(4 * words[idx])[:-1]]).astype(np.int64)
            frs = engine.fragments_from_masks(rec, mask_info[idx], ofs, buf[nrec[r]:nrec[r] + nmask[r]])
            records[idx] = rec
            for j, i in enumerate(idx):
                fragments[i] = frs[j]
        return records, fragments


def deal_images(n_images, world):
    """Image sets (BASELINE.json configs[3]): whole images are dealt to the ranks round-robin; every rank runs the stage on its
    images (GlobalEnergyMinimization.process_many) and the per-image results are gathered once at the end."""
    return [list(range(r, n_images, world)) for r in range(world)]


def gather_objects(local, group=None, dst=0):
    """The one gather at the end of an image-set run: every rank's list of picklable per-image results to rank ``dst``
    (None elsewhere), in rank order."""
    world = dist.get_world_size(group)
    out = [None] * world if dist.get_rank(group) == dst else None
    dist.gather_object(local, out, dst=dst, group=group)
    return out
