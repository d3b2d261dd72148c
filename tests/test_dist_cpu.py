"""The N > 1 path on CPU: world_size 2, gloo.  The collective plumbing (sizes exchange, padded gather, unpack)
and the cost-balanced sharding are device independent."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from superdsm_amd import dist as sdist


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    n = 5 + 3 * rank
    records = torch.from_numpy(rng.integers(0, 255, n * 128, dtype=np.uint8))
    masks = torch.from_numpy(rng.integers(0, 255, 40 + 17 * rank, dtype=np.uint8))
    g = sdist.RecordGather((records, masks), world, rank)
    for _ in range(2):
        g.run()
    if rank == 0:
        parts = g.unpack()
        ok = True
        for r in range(world):
            rr = np.random.default_rng(100 + r)
            exp_rec = rr.integers(0, 255, (5 + 3 * r) * 128, dtype=np.uint8)
            exp_mask = rr.integers(0, 255, 40 + 17 * r, dtype=np.uint8)
            ok &= np.array_equal(parts[r][0], exp_rec) and np.array_equal(parts[r][1], exp_mask)
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() is True


def test_shard_indices_balanced_and_complete():
    rng = np.random.default_rng(1)
    costs = rng.integers(100, 20000, 501)
    for world in (1, 2, 4, 8):
        shards = sdist.shard_indices(costs, world)
        allidx = np.sort(np.concatenate(shards))
        np.testing.assert_array_equal(allidx, np.arange(501))
        tot = np.array([costs[s].sum() for s in shards], float)
        assert tot.max() / tot.mean() < 1.05
