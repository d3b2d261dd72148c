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


def _gpu_solve_local(image, footprints, cfg, mask_info, cache=None):
    """Default local solver of a shard: one engine batch on this rank's GPU.  Returns (records, masks) as uint8 DEVICE tensors
    (nothing is copied to the host here); ``mask_info`` is what every rank expects the shard's mask boxes to be.  ``cache`` (a dict
    owned by a :class:`ShardedBatch`): the plan and workspace are built once and launched again by later steps."""
    from . import _capi, engine
    batch = cache.get('batch') if cache is not None else None
    if batch is None:
        # scheduling as objects._solve: a shard that cannot fill the GPU runs in latency mode (shortest wall clock of ONE batch), a large
        # one in throughput mode (results do not depend on the mode)
        from .objects import LATENCY_MODE_BELOW
        batch = engine.Batch(image, footprints, cfg, mode=1 if len(footprints) < LATENCY_MODE_BELOW else 0)
        assert np.array_equal(batch.mask_info[:len(footprints)], mask_info), 'plan of the shard disagrees with the replicated layout'
        from .objects import _starting_points
        batch.start = _starting_points(batch, cfg)           # callable dsm/init: starting points of this shard's candidates (None otherwise)
        if cache is not None:
            cache['batch'] = batch
    batch.launch()
    n = len(footprints)
    if n:
        # candidates whose workgroup group was given up (a scheduling event on an oversubscribed GPU, SDSM_CAND_GIVEN_UP) are solved
        # again without groups on the owning rank, before the all-gather -- as objects._solve does for unsharded batches
        rec32 = batch.records_dev.view(torch.int32).reshape(-1, 32)
        again = torch.nonzero(rec32[:n, 16] == _capi.CAND_GIVEN_UP).flatten().cpu().numpy()       # (status: int32 #16 of the 128-byte record)
        if again.size:
            sub = engine.Batch(image, [footprints[i] for i in again], cfg, mode=2)
            if batch.start is not None:
                sub.set_start([batch.start[i] for i in again])
            sub.launch()
            sub32 = sub.records_dev.view(torch.int32).reshape(-1, 32)
            for j, i in enumerate(again):
                rec32[i] = sub32[j]
                nb = 4 * ((int(batch.mask_info[i, 2]) * int(batch.mask_info[i, 3]) + 31) // 32)
                o, so = int(batch.mask_offset[i]), int(sub.mask_offset[j])
                batch.masks_dev[o:o + nb] = sub.masks_dev[so:so + nb]
            torch.cuda.current_stream().synchronize()                      # (sub's buffers are released on return)
    return batch.records_dev[:n * 128], batch.masks_dev, batch          # (the batch keeps the buffers alive)


_gpu_solve_local.supports_cache = True


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

    def prepare(self, image, footprints, cfg):
        """Host planning of ONE batch over the ranks (identical on all ranks, no communication): shards by cost, mask boxes and
        payload sizes of every rank, this rank's engine batch.  Returns a :class:`ShardedBatch`; its ``step()`` is launch + the one
        all-gather, ``results()`` unpacks the gathered block."""
        return ShardedBatch(self, image, footprints, cfg)

    def solve(self, image, footprints, cfg):
        sb = self.prepare(image, footprints, cfg)
        sb.step()
        return sb.results()


class ShardedBatch:
    """One batch of candidates dealt to the ranks of a :class:`Sharder` (strong scaling: BASELINE.json configs[4]).  ``step()`` can be
    repeated (a benchmark's timed loop): local solve of this rank's share, pack on the device, ONE all-gather."""

    def __init__(self, sharder, image, footprints, cfg):
        from . import engine
        self.sh = sharder
        self.n = len(footprints)
        self.footprints = [sorted(int(a) for a in fp) for fp in footprints]
        area = np.asarray(image.atom_stats).reshape(-1, 6)[:, 0]
        costs = [sum(int(area[a]) for a in fp if 0 < a < len(area)) for fp in self.footprints]
        self.shards = shard_indices(costs, sharder.world)
        # mask boxes of all candidates and with them every rank's payload size: host-only planning, identical on all ranks
        self.mask_info = engine.plan_mask_boxes(image, self.footprints, cfg)
        self.words = (self.mask_info[:, 2].astype(np.int64) * self.mask_info[:, 3] + 31) // 32
        self.nrec = [128 * len(sh) for sh in self.shards]
        self.nmask = [max(4, int(4 * self.words[sh].sum())) for sh in self.shards]       # (an empty plan still has a 4-byte mask buffer)
        self.pad = max(a + b for a, b in zip(self.nrec, self.nmask))
        self.image, self.cfg = image, cfg
        self.mine = self.shards[sharder.rank]
        self.local = None                                       # what solve_local returned last (keeps its buffers alive)
        self._cache = {}
        self.send = self.recv = self.host = None

    def step(self):
        sh = self.sh
        r = sh.rank
        extra = {'cache': self._cache} if getattr(sh.solve_local, 'supports_cache', False) else {}
        self.local = sh.solve_local(self.image, [self.footprints[i] for i in self.mine], self.cfg, self.mask_info[self.mine], **extra)
        rec_t, mask_t = self.local[0], self.local[1]
        dev = rec_t.device if sh.device is None else torch.device(sh.device)
        if self.send is None:
            self.send = torch.zeros(self.pad, dtype=torch.uint8, device=dev)
            self.recv = torch.empty(sh.world * self.pad, dtype=torch.uint8, device=dev)
        self.send[:self.nrec[r]].copy_(rec_t.reshape(-1)[:self.nrec[r]])
        self.send[self.nrec[r]:self.nrec[r] + self.nmask[r]].copy_(mask_t.reshape(-1)[:self.nmask[r]])
        if dist.get_backend(sh.group) == 'gloo' and self.send.is_cuda:          # rehearsal on one GPU: gloo moves host memory
            hs, hr = self.send.cpu(), torch.empty(sh.world * self.pad, dtype=torch.uint8)
            dist.all_gather(list(hr.chunk(sh.world)), hs, group=sh.group)
            self.host = hr.numpy()
        else:
            dist.all_gather(list(self.recv.chunk(sh.world)), self.send, group=sh.group)   # the one collective of the batch
            self.host = None

    def results(self):
        from . import _capi, engine
        host = self.host if self.host is not None else self.recv.cpu().numpy()
        records = np.zeros(self.n, _capi.RECORD_DTYPE)
        fragments = [None] * self.n
        for r in range(self.sh.world):
            idx = self.shards[r]
            if len(idx) == 0:
                continue
            buf = host[r * self.pad:(r + 1) * self.pad]
            rec = buf[:self.nrec[r]].view(_capi.RECORD_DTYPE)
            ofs = np.concatenate([[0], np.cumsum(4 * self.words[idx])[:-1]]).astype(np.int64)
            frs = engine.fragments_from_masks(rec, self.mask_info[idx], ofs, buf[self.nrec[r]:self.nrec[r] + self.nmask[r]])
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
