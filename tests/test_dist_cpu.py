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


def _fake_solve_local(image, footprints, cfg):
    """Deterministic stand-in for the GPU engine: energy = 10 * sum(labels), a 2x3 mask box with a label-dependent bit
    pattern.  Exercises exactly the byte layout the GPU path produces."""
    from superdsm_amd import _capi
    n = len(footprints)
    rec = np.zeros(n, _capi.RECORD_DTYPE)
    info = np.zeros((n, 4), np.int32)
    off = np.zeros(n, np.int64)
    masks = np.zeros(4 * n, np.uint8)
    for i, fp in enumerate(footprints):
        rec['energy'][i] = 10.0 * sum(fp)
        rec['status'][i] = 0
        rec['n_pixels'][i] = 6
        r0, c0 = 3 + fp[0], 5 + len(fp)
        info[i] = (r0, c0, 2, 3)
        off[i] = 4 * i
        bits = (sum(fp) * 37 + 1) & 0x3f
        masks[4 * i] = bits
        grid = np.array([(bits >> b) & 1 for b in range(6)], bool).reshape(2, 3)
        rr, cc = np.nonzero(grid)
        rec['fg_r0'][i], rec['fg_c0'][i] = r0 + rr.min(), c0 + cc.min()
        rec['fg_h'][i], rec['fg_w'][i] = rr.max() - rr.min() + 1, cc.max() - cc.min() + 1
    return rec.view(np.uint8).reshape(-1), info, off, masks


def _shard_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    fps = [[a] for a in range(1, 12)] + [[a, a + 1] for a in range(1, 8)] + [[2, 3, 4]]
    sh = sdist.Sharder(device='cpu', solve_local=_fake_solve_local)
    recs, frags = sh.solve(object(), fps, {})
    solo_rec, info, off, masks = _fake_solve_local(None, fps, {})
    from superdsm_amd import _capi
    solo = solo_rec.view(_capi.RECORD_DTYPE)
    solo_frags = sdist.fragments_from_masks(solo, info, off, masks)
    ok = np.array_equal(recs['energy'], solo['energy'])
    for a, b in zip(frags, solo_frags):
        ok &= np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_batch_equals_single_rank_world2_gloo():
    """Every rank ends up with the full, correctly ordered result of the batch (records and mask fragments)."""
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = dict(q.get() for _ in range(2))
    assert got == {0: True, 1: True}
