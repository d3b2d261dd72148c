"""The N > 1 path on CPU: world_size 2, gloo.  The collective plumbing (padded gather / all-gather, unpack), the cost-balanced
sharding, the image-set dealing and bench.py's own rank start-up are device independent."""
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


class _FakeImage:
    """What a rank needs to know about the (replicated) image to plan a batch: shape and per-atom statistics."""
    H, W, n_atoms, background_margin = 64, 80, 12, 4.0

    def __init__(self):
        st = np.zeros((self.n_atoms + 1, 6), np.int32)
        for l in range(1, self.n_atoms + 1):
            r0, c0 = 3 + 4 * (l % 5), 2 + 5 * l
            st[l] = (6 + l, r0, r0 + 2 + l % 3, c0, c0 + 3 + l % 2, 0)       # area, rmin, rmax, cmin, cmax
        self.atom_stats = st.reshape(-1)


_CFG = dict(scale=1000, epsilon=1.0, alpha=0.033, smooth_amount=4, smooth_subsample=8, gaussian_shape_multiplier=2)


def _fake_solve_local(image, footprints, cfg, mask_info):
    """Deterministic stand-in for the GPU engine on exactly the byte layout the GPU path produces: 128-byte records and
    bit-packed masks over the PLANNED region boxes (a label-dependent bit pattern), as uint8 tensors."""
    from superdsm_amd import _capi
    n = len(footprints)
    rec = np.zeros(n, _capi.RECORD_DTYPE)
    words = (mask_info[:, 2].astype(np.int64) * mask_info[:, 3] + 31) // 32
    masks = np.zeros(max(1, int(words.sum())), np.uint32)
    pos = 0
    for i, fp in enumerate(footprints):
        r0, c0, h, w = (int(v) for v in mask_info[i])
        rng = np.random.default_rng(sum(fp) * 131 + len(fp))
        grid = rng.random((h, w)) < 0.4
        grid[h // 2, w // 2] = True
        flat = np.flatnonzero(grid.reshape(-1))
        np.bitwise_or.at(masks, pos + flat // 32, np.uint32(1) << (flat % 32).astype(np.uint32))
        rr, cc = np.nonzero(grid)
        rec['energy'][i] = 10.0 * sum(fp)
        rec['n_pixels'][i] = h * w
        rec['fg_r0'][i], rec['fg_c0'][i] = r0 + rr.min(), c0 + cc.min()
        rec['fg_h'][i], rec['fg_w'][i] = rr.max() - rr.min() + 1, cc.max() - cc.min() + 1
        pos += int(words[i])
    return torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()), torch.from_numpy(masks.view(np.uint8).copy())


def _shard_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from superdsm_amd import _capi, engine
    fps = [[a] for a in range(1, 12)] + [[a, a + 1] for a in range(1, 8)] + [[2, 3, 4]]
    img = _FakeImage()
    sh = sdist.Sharder(device='cpu', solve_local=_fake_solve_local)
    calls = []
    real_gather = dist.all_gather
    dist.all_gather = lambda *a, **k: calls.append(1) or real_gather(*a, **k)
    recs, frags = sh.solve(img, fps, _CFG)
    dist.all_gather = real_gather
    info = engine.plan_mask_boxes(img, fps, _CFG)
    solo_rec, solo_masks = _fake_solve_local(img, fps, _CFG, info)
    solo = solo_rec.numpy().view(_capi.RECORD_DTYPE)
    words = (info[:, 2].astype(np.int64) * info[:, 3] + 31) // 32
    off = np.concatenate([[0], np.cumsum(4 * words)[:-1]]).astype(np.int64)
    solo_frags = sdist.fragments_from_masks(solo, info, off, solo_masks.numpy())
    ok = np.array_equal(recs['energy'], solo['energy']) and len(calls) == 1          # ONE collective per batch, no size exchange
    for a, b in zip(frags, solo_frags):
        ok &= np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # image sets: whole images dealt to ranks, one gather of the per-image results at the end
    mine = sdist.deal_images(5, world)[rank]
    got = sdist.gather_objects([(i, f'result of image {i}') for i in mine], dst=0)
    if rank == 0:
        flat = sorted(x for part in got for x in part)
        ok &= flat == [(i, f'result of image {i}') for i in range(5)]
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_batch_equals_single_rank_world2_gloo():
    """Every rank ends up with the full, correctly ordered result of the batch (records and mask fragments) after exactly one
    collective; image sets are dealt by image and gathered once."""
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


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts two child ranks itself (rendezvous on 127.0.0.1, gloo on a box
    without GPUs) and prints ONE JSON line with n_gpus == 2; --dry-run skips the GPU work."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run'],
                         env=env, capture_output=True, timeout=180)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['ms_per_step'] == 2.0      # MAX over the ranks' (1 + rank)


def test_bench_ranks_per_gpu_starts_that_many_processes_per_card():
    """`--ranks-per-gpu R`: R worker processes per GPU (an image set is host bound per process); n_gpus stays the number of GPUs."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--ranks-per-gpu', '3', '--dry-run'],
                         env=env, capture_output=True, timeout=180)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 1 and out['ms_per_step'] == 3.0 and out['config']['backend'] == 'gloo'      # MAX over the three ranks' (1 + rank)


def test_bench_sharded_mode_deals_one_batch_and_gathers_once():
    """`python bench.py --gpus 2 --mode sharded --dry-run`: the strong-scaling entry point of BASELINE.json configs[4] -- ONE batch dealt
    to the ranks by cost (dist.shard_indices), one all-gather of the per-rank payloads -- rehearsed without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--mode', 'sharded', '--dry-run'],
                         env=env, capture_output=True, timeout=180)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['scaling'] == 'strong' and out['config']['mode'] == 'sharded' and out['config']['ranks'] == 2
    assert sum(out['config']['shard_sizes']) == 1000 and max(out['config']['shard_sizes']) - min(out['config']['shard_sizes']) <= 1
    assert out['config']['devices_visible'] == 0


# ---------------------------------------------------------------------------------------------------------
# round 4: more ranks than the one box ever has GPUs -- the first real 1 -> 8 run should hold no surprise in the plumbing
# ---------------------------------------------------------------------------------------------------------
def _bench_dry(*flags, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--dry-run'] + list(flags), env=env, capture_output=True, timeout=timeout)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_sharded_dry_run_world_4_and_8():
    """`--mode sharded --dry-run` with 4 and 8 self-started ranks (gloo): n_gpus, backend, the shards of the one batch (complete, balanced
    to one candidate), the payload every rank contributes to the one all-gather."""
    for world in (4, 8):
        out = _bench_dry('--gpus', str(world), '--mode', 'sharded')
        cfg = out['config']
        assert out['n_gpus'] == world and out['scaling'] == 'strong' and cfg['backend'] == 'gloo' and cfg['ranks'] == world
        assert sum(cfg['shard_sizes']) == 1000 and max(cfg['shard_sizes']) - min(cfg['shard_sizes']) <= 1 and len(cfg['shard_sizes']) == world
        assert cfg['payload_bytes_per_rank'] == 128 * max(cfg['shard_sizes'])
        assert out['ms_per_step'] == float(world)                                  # MAX over the ranks' (1 + rank)


def test_bench_image_set_dry_run_deals_unevenly_and_leaves_ranks_empty():
    """`--mode image_set --dry-run`: 49 images over 8 ranks (7 / 6 per rank), 5 images over 8 ranks (three ranks get none), one gather of
    the per-image results at the end."""
    out = _bench_dry('--gpus', '8', '--mode', 'image_set')
    assert out['config']['images_per_rank'] == [7, 6, 6, 6, 6, 6, 6, 6] and out['n_gpus'] == 8 and out['config']['backend'] == 'gloo'
    out = _bench_dry('--gpus', '8', '--mode', 'image_set', '--images', '5')
    assert out['config']['images_per_rank'] == [1, 1, 1, 1, 1, 0, 0, 0]
    out = _bench_dry('--gpus', '4', '--ranks-per-gpu', '2', '--mode', 'image_set', '--images', '9')
    assert out['n_gpus'] == 4 and out['config']['ranks'] == 8 and out['config']['ranks_per_gpu'] == 2 and sum(out['config']['images_per_rank']) == 9


_GIVEN_UP_CALLS = {}


def _solve_local_with_a_given_up_group(image, footprints, cfg, mask_info):
    """The fake engine; rank 1's first answer reports one of its candidates as SDSM_CAND_GIVEN_UP and solves it again before returning --
    what _gpu_solve_local does with a workgroup group that was given up on an oversubscribed card (the retry is rank-local: before the gather)."""
    from superdsm_amd import _capi
    rec_t, mask_t = _fake_solve_local(image, footprints, cfg, mask_info)
    if dist.get_rank() == 1 and len(footprints) > 0 and not _GIVEN_UP_CALLS.get('done'):
        _GIVEN_UP_CALLS['done'] = True
        rec = rec_t.numpy().view(_capi.RECORD_DTYPE)
        good = rec[0].copy()
        rec['status'][0] = _capi.CAND_GIVEN_UP                    # first attempt
        rec['energy'][0] = np.nan
        again = np.flatnonzero(rec['status'] == _capi.CAND_GIVEN_UP)
        sub_rec, _ = _fake_solve_local(image, [footprints[i] for i in again], cfg, mask_info[again])
        rec[again] = sub_rec.numpy().view(_capi.RECORD_DTYPE)     # second attempt, same rank
        assert rec[0].tobytes() == good.tobytes()
    return rec_t, mask_t


def _shard_worker_many(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from superdsm_amd import _capi, engine
    img = _FakeImage()
    ok = True
    # 19 candidates over 4 ranks (uneven), 3 candidates over 4 ranks (an empty shard), 0 candidates
    for fps in ([[a] for a in range(1, 12)] + [[a, a + 1] for a in range(1, 8)] + [[2, 3, 4]], [[1], [2, 3], [4]], []):
        sh = sdist.Sharder(device='cpu', solve_local=_solve_local_with_a_given_up_group)
        if len(fps) == 0:
            continue                                               # (an empty batch is not sent to the engine: objects.compute_objects returns at once)
        sb = sh.prepare(img, fps, _CFG)
        assert sorted(np.concatenate(sb.shards).tolist()) == list(range(len(fps)))
        if len(fps) < world:
            assert min(len(s) for s in sb.shards) == 0
        calls = []
        real_gather = dist.all_gather
        dist.all_gather = lambda *a, **k: calls.append(1) or real_gather(*a, **k)
        sb.step()
        dist.all_gather = real_gather
        recs, frags = sb.results()
        info = engine.plan_mask_boxes(img, fps, _CFG)
        solo_rec, solo_masks = _fake_solve_local(img, fps, _CFG, info)
        solo = solo_rec.numpy().view(_capi.RECORD_DTYPE)
        words = (info[:, 2].astype(np.int64) * info[:, 3] + 31) // 32
        off = np.concatenate([[0], np.cumsum(4 * words)[:-1]]).astype(np.int64)
        solo_frags = sdist.fragments_from_masks(solo, info, off, solo_masks.numpy())
        ok &= recs.tobytes() == solo.tobytes() and len(calls) == 1 and (recs['status'] != _capi.CAND_GIVEN_UP).all()
        for a, b in zip(frags, solo_frags):
            ok &= np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_batch_world4_uneven_and_empty_shards_and_a_given_up_retry():
    """Four ranks (gloo): 19 candidates (shards of 5 / 5 / 5 / 4), 3 candidates (one rank's shard is empty), a rank whose first local
    answer held a given-up workgroup group (solved again on that rank before the gather): every rank ends with the bytes of the unsharded
    batch after exactly one collective."""
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker_many, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = dict(q.get() for _ in range(4))
    assert got == {0: True, 1: True, 2: True, 3: True}
