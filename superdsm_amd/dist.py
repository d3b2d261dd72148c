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


def _gpu_solve_local(image, footprints, cfg):
    """Default local solver of a shard: one engine batch on this rank's GPU.  Returns (records bytes, mask_info
    int32 [n,4], mask_offset int64 [n], masks bytes), all numpy."""
    from . import engine
    batch = engine.Batch(image, footprints, cfg, latency_mode=True)
    batch.launch()
    torch.cuda.synchronize(image.device)
    n = len(footprints)
    return (batch.records_dev.cpu().numpy()[:n * 128].copy(), batch.mask_info[:n].copy(), batch.mask_offset[:n].copy(),
            batch.masks_dev.cpu().numpy().copy())


def fragments_from_masks(records, mask_info, mask_offset, masks):
    """Foreground fragments from bit-packed region-bbox masks (same rule as engine.Batch.fragments)."""
    from . import _capi
    out = []
    for i, r in enumerate(records):
        if r['fg_h'] <= 0 or r['status'] in (_capi.CAND_TRIVIAL, _capi.CAND_ERROR):
            out.append((np.zeros(2, int), np.zeros((1, 1), bool)))
            continue
        r0, c0, h, w = (int(v) for v in mask_info[i])
        nbytes = ((h * w + 31) // 32) * 4
        bits = np.unpackbits(masks[mask_offset[i]:mask_offset[i] + nbytes], bitorder='little')[:h * w].reshape(h, w)
        fr, fc = int(r['fg_r0']) - r0, int(r['fg_c0']) - c0
        out.append((np.array([int(r['fg_r0']), int(r['fg_c0'])]), bits[fr:fr + int(r['fg_h']), fc:fc + int(r['fg_w'])].astype(bool)))
    return out


class Sharder:
    """Splits every batch of candidates over the ranks of a process group and all-gathers the results, so that every
    rank can continue the (cheap, deterministic) host-side generation logic in lock step.

    Per batch: each rank solves its cost-balanced share on its own GPU (no data-path collective), then ONE
    all-gather of a padded byte block per rank: records (128 B / candidate), mask boxes and the bit-packed masks.
    """

    def __init__(self, group=None, device=None, solve_local=_gpu_solve_local):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device
        self.solve_local = solve_local

    def solve(self, image, footprints, cfg):
        from . import _capi
        n = len(footprints)
        stats = getattr(image, 'atom_stats', None)
        if stats is not None:
            area = np.asarray(stats).reshape(-1, 6)[:, 0]
            costs = [sum(int(area[a]) for a in fp if 0 < a < len(area)) for fp in footprints]
        else:
            costs = [len(fp) for fp in footprints]
        shards = shard_indices(costs, self.world)
        mine = shards[self.rank]
        rec_b, info, off, masks = self.solve_local(image, [footprints[i] for i in mine], cfg)
        payload = np.concatenate([np.asarray(rec_b, np.uint8).reshape(-1), np.ascontiguousarray(info, np.int32).view(np.uint8).reshape(-1),
                                  np.ascontiguousarray(off, np.int64).view(np.uint8).reshape(-1), np.asarray(masks, np.uint8).reshape(-1)])
        dev = self.device if self.device is not None else (image.device if hasattr(image, 'device') else 'cpu')
        size = torch.tensor([payload.size], dtype=torch.int64, device=dev)
        sizes = [torch.zeros_like(size) for _ in range(self.world)]
        dist.all_gather(sizes, size, group=self.group)
        sizes = [int(s.item()) for s in sizes]
        pad = max(sizes)
        send = torch.zeros(pad, dtype=torch.uint8, device=dev)
        send[:payload.size] = torch.from_numpy(payload).to(dev)
        recv = [torch.zeros(pad, dtype=torch.uint8, device=dev) for _ in range(self.world)]
        dist.all_gather(recv, send, group=self.group)                 # the one data collective of the batch
        records = np.zeros(n, _capi.RECORD_DTYPE)
        fragments = [None] * n
        for r in range(self.world):
            idx = shards[r]
            k = len(idx)
            buf = recv[r].cpu().numpy()[:sizes[r]]
            rec = buf[:k * 128].view(_capi.RECORD_DTYPE)
            inf = buf[k * 128:k * 144].view(np.int32).reshape(k, 4)
            ofs = buf[k * 144:k * 152].view(np.int64)
            msk = buf[k * 152:]
            frs = fragments_from_masks(rec, inf, ofs, msk)
            for j, i in enumerate(idx):
                records[i] = rec[j]
                fragments[i] = frs[j]
        return records, fragments
