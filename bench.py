#!/usr/bin/env python3
"""Headline benchmark: candidate DSM solves per second on the BBBC039-like 520x696 image (BASELINE.json configs[1]),
1..N MI355X (one process per GPU).

A "step" = one pass of the hot path over one batch of synthetic input with everything already resident in HBM: ONE
launch of the engine over the complete candidate lists of `--images` (default 8) BBBC039-like images -- region crops, greedy
grids + G~ rows, elliptical + DSM solves, masks, records (one sdsm_batch_launch_multi) -- plus, for N > 1, the single gather of
the fixed-size records and bit-packed masks to rank 0 over RCCL.  One image alone (501 candidates) cannot fill 256 compute
units; the multi-image plan is the production answer to that (an image set, or the same generation of several images), it needs
neither several streams nor an environment variable.

`python bench.py --gpus N` with WORLD_SIZE unset starts the N ranks itself (fresh child processes, created before anything
touches the GPU); under torchrun it is one rank.  Prints ONE JSON line on rank 0.

The 8 images of a step are 8 DIFFERENT BBBC039-like images (ellipses at the centres / areas of eight of the reference's per-image
regression tables, 68 .. 170 objects).  The default run also measures the other BASELINE.json configs (`extras.configs`: GOWT1-like
frame, NIH3T3-like image set through the stage, synthetic 4096^2), each with its launch time, solves/s and roofline fraction.

Other workloads: --workload {gowt1_like,nih3t3_like,synthetic4096,synthetic512,synthetic256} (one image per step unless --images);
--mode image_set: BASELINE.json configs[3] -- a set of NIH3T3-like images dealt to the ranks, every rank runs the
global-energy-minimisation stage on its images in lock step (process_many), one gather of the results at the end;
--mode sharded: BASELINE.json configs[4] as specified -- the candidates of ONE batch (default: the synthetic 4096^2 image) dealt to the
ranks by cost, every rank solves its share, ONE all-gather of records + masks per step (strong scaling).
"""
import argparse
import gc
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='bbbc039_like')
    ap.add_argument('--mode', default='solves', choices=['solves', 'image_set', 'sharded'])
    ap.add_argument('--min-gpu-seconds', type=float, default=2.0, help='the timed regions are repeated until they add up to this much (so that an outside sampler of GPU activity sees the run)')
    ap.add_argument('--no-configs', action='store_true', help='skip the other BASELINE.json configs in extras')
    ap.add_argument('--same-layout', action='store_true', help='the step of rounds 1-2: 8 copies of ONE BBBC039-like image (layout 0) instead of 8 different ones')
    ap.add_argument('--images', type=int, default=None, help='images per plan = per step (default: 8 for bbbc039_like / synthetic256, else 1)')
    ap.add_argument('--max-size', type=int, default=3, help='candidates = connected atom subsets up to this size + universes')
    ap.add_argument('--repeats', type=int, default=5, help='the timed region of --steps steps is repeated; the median is reported')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of each CPU baseline variant')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the stage / compute_objects / preprocessing timings')
    ap.add_argument('--no-ranks-per-gpu', action='store_true', help='skip the child runs of the stage with 1 / 2 / 4 worker processes on the card (extras)')
    ap.add_argument('--inflight', type=int, default=1, help='independent steps in flight on separate streams (1 = strictly sequential launches)')
    ap.add_argument('--images-workload-explicit', action='store_true', help='image_set mode: take --workload literally (default: bbbc039_like means the NIH3T3-like set of BASELINE.json configs[3])')
    ap.add_argument('--ranks-per-gpu', type=int, default=1, help='worker processes per GPU (image sets are host bound per process: several processes share a card; gloo moves the small results)')
    ap.add_argument('--dry-run', action='store_true', help='rendezvous and reporting only, no GPU work (CPU test of the N-rank plumbing)')
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------------
# N ranks from one command line
# ---------------------------------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """Parent of `python bench.py --gpus N`: starts N fresh child processes (nothing here has touched the GPU), one rank each,
    passes rank 0's JSON line through and returns the largest exit code."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus * args.ranks_per_gpu):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus * args.ranks_per_gpu), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = max(rc, p.wait())
    for line in out0.decode().splitlines():                   # the JSON line to stdout, anything a library printed (gloo's connection notes) to stderr
        (sys.stdout if line.startswith('{') else sys.stderr).write(line + '\n')
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = CPU restatement of the reference path, kind "port")
# ---------------------------------------------------------------------------------------------------------------------------
def cpu_baseline(scenes, budget_s):
    """The oracle on the same work as one GPU step -- the candidates of all images, every image's in ONE OpenMP loop, one candidate
    per worker (Ray: one task per core, objects.py:280) -- in two variants: a worker per core, and cores / 2 workers with 2 threads
    each (MKL_NUM_THREADS: 2 of the reference's task specs, examples/BBBC039/task.json:4).  Each variant is a bounded sample."""
    from oracle import oracle
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    uniq = list({id(sc): sc for sc in scenes}.values())
    mult = len(scenes) // len(uniq)
    ncand = sum(len(sc['footprints']) for sc in scenes)

    def one_pass(workers, inner):
        for sc in uniq:
            oracle.compute_objects(sc['y'], None, sc['atoms'], sc['footprints'] * mult, sc['dsm_cfg'], nthreads=workers, inner_threads=inner)

    out = {}
    for key, workers, inner in (('per_core', cores, 1), ('half_cores_x2', max(1, cores // 2), 2)):
        t0 = time.time()
        one_pass(workers, inner)
        t_once = time.time() - t0
        reps = int(max(1, min(50, round(budget_s / max(t_once, 1e-3)))))
        t0 = time.time()
        for _ in range(reps):
            one_pass(workers, inner)
        dt = time.time() - t0
        out[key] = dict(value=reps * ncand / dt, workers=workers, threads_per_worker=inner,
                        sample=f'{ncand} candidates (the {len(scenes)} images of one GPU step, one OpenMP loop per image) x {reps} passes, {dt:.1f} s wall')
    best = max(out.values(), key=lambda v: v['value'])
    return dict(value=best['value'], unit='candidate solves/s', cores=cores, kind='port', sample=best['sample'],
                variants=out, note='CPU restatement of the reference path (oracle/), not the reference itself and not the target')


def cpu_stage_wall(scene, beta, pruning):
    """GlobalEnergyMinimization.process with every batch solved by the oracle on all host cores: the CPU wall clock per image."""
    from oracle import oracle
    from superdsm_amd import config, globalenergymin
    import unittest.mock as mock

    def oracle_compute(objs, y, atoms, dsm_cfg, log_root_dir, status_line=None, out=None, shard=None):
        objs = list(objs)
        if not objs:
            return
        recs, frags, _ = oracle.compute_objects(y.model, None, atoms, [sorted(o.footprint) for o in objs], dsm_cfg, nthreads=0)
        for o, r, f in zip(objs, recs, frags):
            o.energy, o.is_optimal, o.on_boundary, o.processing_time = float(r['energy']), bool(r['is_optimal']), bool(r['on_boundary']), 0
            o.fg_offset, o.fg_fragment = np.array(r['fg_offset']), f

    stage = globalenergymin.GlobalEnergyMinimization()
    data = dict(y=scene['y'], y_mask=np.ones(scene['y'].shape, bool), atoms=scene['atoms'], adjacencies=scene['adjacencies'], dsm_cfg=scene['dsm_cfg'])
    cfg = config.Config({'global-energy-minimization': {'beta': beta, 'pruning': pruning}})
    with mock.patch.object(globalenergymin, 'compute_objects', oracle_compute):
        t0 = time.perf_counter()
        stage(data, cfg, out='muted')
        dt = time.perf_counter() - t0
    return dt * 1e3, data['performance'].overall_computed_object_count


def source_hash():
    h = hashlib.sha1()
    d = os.path.join(ROOT, 'superdsm_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def measured_traffic(workload, n_images):
    """HBM bytes per launch of the solve kernels from the PMC passes of THIS round's sources (profiles/r*_pmc_summary.json, written
    by tools/summarize_profiles.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command): only if the summary was
    taken from the kernel sources that are running now, else None."""
    best = None
    for f in sorted(os.listdir(os.path.join(ROOT, 'profiles'))):
        if f.endswith('_pmc_summary.json'):
            try:
                d = json.load(open(os.path.join(ROOT, 'profiles', f)))
            except Exception:
                continue
            if d.get('source_hash') == source_hash() and d.get('workload') == workload and d.get('images_per_launch') == n_images:
                best = d.get('solve_hbm_bytes_per_launch')
    return best


# ---------------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.gpus * args.ranks_per_gpu > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '16' if args.inflight > 4 else '8')      # ROCm maps streams onto 4 hardware queues by default; a launch uses the caller's stream + 3
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    ndev = torch.cuda.device_count()                           # (does not initialise the GPU)
    # fewer GPUs than ranks (rehearsal on a one-GPU box) or no GPU at all (--dry-run on a CPU box): gloo moves host memory
    rpg = max(1, args.ranks_per_gpu)
    backend = os.environ.get('SDSM_BENCH_BACKEND', 'nccl' if ndev * rpg >= world and rpg == 1 and not args.dry_run else 'gloo')
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend, rank=rank, world_size=world)
    # GPUs this run really uses: ranks are placed on device (local_rank // rpg) % ndev -- fewer devices than ranks / rpg means sharing
    # (a dry run on a box without devices reports the requested count next to devices_visible = 0)
    n_gpus_used = max(1, min(max(1, world // rpg), ndev)) if ndev > 0 else max(1, world // rpg)
    if args.dry_run:
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        shard_sizes = payload = images_per_rank = None
        if world > 1:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if args.mode == 'image_set':                              # the dealing of an image set and its one gather of per-image results
            from superdsm_amd import dist as sdist
            n_img = args.images if args.images else 49            # (BASELINE.json configs[3]: the NIH3T3 set has 49 images)
            deal = sdist.deal_images(n_img, world)
            local = [(i, f'objects of image {i}') for i in deal[rank]]
            got = sdist.gather_objects(local, dst=0) if world > 1 else [local]
            if rank == 0:
                assert sorted(x for part in got for x in part) == [(i, f'objects of image {i}') for i in range(n_img)]
            images_per_rank = [len(d) for d in deal]
        if args.mode == 'sharded':                                # the dealing of one batch and its one all-gather, on fake costs / payloads
            from superdsm_amd import dist as sdist
            costs = np.random.default_rng(5).integers(200, 20000, 1000)
            shards = sdist.shard_indices(costs, world)
            pad = max(128 * len(sh) for sh in shards)
            send = torch.full((pad,), rank, dtype=torch.uint8)
            recv = torch.empty(world * pad, dtype=torch.uint8)
            if world > 1:
                dist.all_gather(list(recv.chunk(world)), send)
            else:
                recv.copy_(send)
            assert all(int(recv[r * pad]) == r for r in range(world))
            shard_sizes = [len(sh) for sh in shards]
            payload = int(pad)
        if rank == 0:
            print(json.dumps({'metric': 'candidate DSM solves/sec', 'value': 0.0, 'unit': 'candidate solves/s', 'n_gpus': n_gpus_used, 'steps': args.steps,
                              'warmup': args.warmup, 'ms_per_step': float(t.item()), 'higher_is_better': True, 'scaling': 'strong' if args.mode == 'sharded' else 'weak', 'vs_baseline': None,
                              'dtype': 'f64', 'data': 'none (dry run of the rank plumbing)',
                              'config': {'workload': 'dry-run', 'backend': backend, 'mode': args.mode, 'ranks': world, 'ranks_per_gpu': rpg, 'devices_visible': ndev, 'shard_sizes': shard_sizes,
                                         'payload_bytes_per_rank': payload, 'images_per_rank': images_per_rank}}))
        if world > 1:
            dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the DSM solve path has no CPU fallback)'
    torch.cuda.set_device((local_rank // rpg) % ndev)
    if args.mode == 'image_set':
        return image_set_mode(args, world, rank, backend, n_gpus_used)
    if args.mode == 'sharded':
        return sharded_mode(args, world, rank, backend, n_gpus_used)

    from superdsm_amd import _capi, engine, testing
    from superdsm_amd import dist as sdist

    n_images = args.images if args.images else (8 if args.workload in ('bbbc039_like', 'synthetic256') else 1)
    # `n_images` images per step; BBBC039-like: DIFFERENT images (eight of the reference's per-image object tables place the nuclei)
    scenes = [testing.make_scene(args.workload, max_size=args.max_size, layout_index=k % 8 if args.workload == 'bbbc039_like' else 0) for k in range(n_images)] \
        if args.workload == 'bbbc039_like' and not args.same_layout else [testing.make_scene(args.workload, max_size=args.max_size)] * n_images
    scene = scenes[0]
    fps1 = scene['footprints']
    margin = scene['dsm_cfg']['background_margin']
    imgs = [engine.DeviceImage(sc['y'], None, sc['atoms'], margin) for sc in scenes]
    fps = [fp for sc in scenes for fp in sc['footprints']]
    image_of = np.concatenate([np.full(len(sc['footprints']), k, np.int32) for k, sc in enumerate(scenes)])
    nfl = max(1, min(args.inflight, args.steps))
    batches = [engine.Batch(imgs, fps, scene['dsm_cfg'], image_of=image_of) for _ in range(nfl)]
    streams = [torch.cuda.Stream() for _ in range(nfl)]
    sizes = [[batches[0].records_dev.numel(), batches[0].masks_dev.numel()]] * world       # every rank has the same plan: nothing to exchange
    gathers = [sdist.RecordGather(b, world, rank, sizes=sizes) if world > 1 else None for b in batches]
    batch = batches[0]
    L = _capi.lib()

    def step(i):
        k = i % nfl
        if gathers[k] is not None:            # slot k's previous gather (default stream) must have read its buffers
            streams[k].wait_stream(torch.cuda.default_stream())
        with torch.cuda.stream(streams[k]):
            batches[k].launch()
        if gathers[k] is not None:            # the gather is queued behind this step's kernels; the host does not block
            torch.cuda.default_stream().wait_stream(streams[k])
            gathers[k].run()

    for k in range(nfl):                       # untimed: first launch of every plan (kernel attributes, side streams)
        step(k)
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    region_ms = []
    n_regions = max(1, args.repeats)
    while len(region_ms) < n_regions:          # the timed region: EXACTLY --steps steps, barrier + synchronize on both sides
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        region_ms.append(dt * 1e3)
        if len(region_ms) == n_regions and sum(region_ms) < args.min_gpu_seconds * 1e3 and n_regions < 400:
            n_regions = min(400, max(n_regions + 1, int(np.ceil(args.min_gpu_seconds * 1e3 / np.median(region_ms)))))   # (the same count on every rank: the times are all-reduced)
    dt = float(np.median(region_ms)) * 1e-3

    # kernel-level timing with HIP events on the launch stream (one launch at a time)
    L.sdsm_enable_kernel_timing(1)
    solve_ms, setup_ms = [], []
    for _ in range(5):
        batch.launch()
        solve_ms.append(L.sdsm_last_solve_kernel_ms())
        setup_ms.append(L.sdsm_last_setup_kernel_ms())
    L.sdsm_enable_kernel_timing(0)
    recs = batch.records()
    n_total = len(fps) * world
    value = n_total * args.steps / dt
    alg_bytes = engine.algorithmic_bytes(recs, batch.mask_info)
    kern_ms = float(np.mean(solve_ms))
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    evals = recs['evals_value'].astype(np.int64) + recs['evals_full'].astype(np.int64)
    # FP64-vector cross-check of the Hessian phase: flops_c = E_c N_c (4 (6 + z) + 20) + H_c N_c (6 + z)^2, z ~ 11
    z = 11.0
    flops = float((evals * recs['n_pixels'] * (4 * (6 + z) + 20)).sum() + (recs['evals_full'].astype(np.int64) * recs['n_pixels'] * (6 + z) ** 2).sum())
    out = {
        'metric': 'candidate DSM solves/sec', 'value': value, 'unit': 'candidate solves/s', 'n_gpus': n_gpus_used, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'{args.workload} {scene["y"].shape[0]}x{scene["y"].shape[1]} (BASELINE.json configs[1] stand-in: ellipses at the centres/areas of '
                               f'{len(set(id(sc) for sc in scenes))} different reference BBBC039 regression tables), {n_images} images per step in ONE launch, all candidates of every image: connected atom '
                               f'subsets of size <= {args.max_size} + cluster universes',
                   'images_per_step': n_images, 'candidates_per_image': [len(sc['footprints']) for sc in scenes], 'candidates_per_step_per_gpu': len(fps),
                   'atoms_per_image': [int(sc['atoms'].max()) for sc in scenes], 'backend': backend if world > 1 else None,
                   'candidates_per_rank': [len(fps)] * world, 'gather_payload_bytes_per_rank': int(sum(sizes[0])) if world > 1 else 0,
                   'ranks': world, 'ranks_per_gpu': rpg, 'devices_visible': ndev,
                   'median_N': int(np.median(recs['n_pixels'])), 'median_M': int(np.median(recs['n_deform'])),
                   'parallelism': f'{world} ranks ({rpg} per GPU), weak scaling: every rank solves its own images, one gather of records + masks per step (RCCL with one rank per GPU)'
                                  if world > 1 else 'single GPU',
                   'steps_in_flight': nfl, 'timed_regions': len(region_ms), 'timed_regions_ms_min_median_max': [float(np.min(region_ms)), float(np.median(region_ms)), float(np.max(region_ms))],
                   'timed_seconds_total': float(sum(region_ms) * 1e-3), 'value_is': 'median over the timed regions'},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s', 'frac': achieved / 8000.0,
                     'traffic': measured_traffic(args.workload, n_images),
                     'kernel': 'sdsm_k_solve: the size classes of one launch, which run concurrently on four queues (class 1 <128, 2560, .., 192> holds '
                               f'{int(((recs["n_deform"] + 6 <= 128)).sum())} of the {len(recs)} candidates; the longest chains are the workgroup groups of the 10-17 k-pixel clusters and class 1b); '
                               'kernel_ms = HIP events on the launch stream from the end of sdsm_k_setup to the join of the classes (includes sdsm_k_setup_rows, the rows of G~ of the '
                               'regions above 4096 pixels); per-class rocprofv3 averages: profiles/r04_kernel_stats.csv',
                     'kernel_ms': kern_ms, 'setup_kernel_ms': float(np.mean(setup_ms)), 'algorithmic_bytes_per_launch': alg_bytes,
                     'achieved_in_timed_region': alg_bytes / (dt / args.steps) / 1e9,
                     'fp64_vector_tflops': flops / (kern_ms * 1e-3) / 1e12, 'fp64_vector_frac_of_78.6': flops / (kern_ms * 1e-3) / 1e12 / 78.6,
                     'pixel_evaluations_per_launch': int((evals * recs['n_pixels']).sum()),
                     'pixel_evaluations_per_image': int((evals * recs['n_pixels']).sum() // n_images),
                     'algorithmic_bytes_definition': 'SURVEY.md 8(d): sum over candidates of E_c (12 N_c + 8 (6 + M_c)) + 12 N_c + ceil(bbox_c / 8) + 128, E_c = passes over the pixels actually made'},
        'status_counts': {str(k): int(v) for k, v in zip(*np.unique(recs['status'], return_counts=True))},
        # candidates whose region is separable or all but (psi below 1e-4 of the empty model's N ln 2: "no energy"): status optimal, but inf psi = 0 is not attained and the value is the stopping
        # rule's -- what cvxopt would report there is unpinned; they never win a set cover (beta >= 100)
        'near_separable_candidates': int((recs['energy'] < 1e-4 * recs['n_pixels'] * np.log(2)).sum()),
    }
    if rank == 0 and world == 1 and not args.no_extras:
        out['extras'] = extras(args, scene, imgs[0], n_images, scenes)
        if not args.no_configs and args.workload == 'bbbc039_like':
            out['extras']['configs'] = other_configs(args)
    if rank == 0 and not args.no_cpu and world == 1:
        out['cpu_baseline'] = cpu_baseline(scenes, args.cpu_seconds)
        if 'extras' in out and args.workload in ('bbbc039_like', 'synthetic256'):
            ms, ncomp = cpu_stage_wall(scene, 150.0, 'isbi24')
            out['extras']['stage_wall_ms_per_image_cpu_oracle'] = ms
            out['extras']['stage_candidates_per_image'] = ncomp
    elif rank == 0:
        out['cpu_baseline'] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _capi_lib():
    from superdsm_amd import _capi
    return _capi.lib()


def extras(args, scene, img, n_images, scenes=None):
    """What BASELINE.json's metric names beside the solver throughput: wall clock per image through the product entry points."""
    import torch
    from superdsm_amd import config, engine, globalenergymin, image, objects
    ex = {}
    # (1) one image alone, one launch (latency scheduling), device only
    lat = engine.Batch(img, scene['footprints'], scene['dsm_cfg'], latency_mode=True)
    lat.launch()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t1 = time.perf_counter()
        lat.launch()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t1) * 1e3)
    ex['single_image_launch_ms'] = float(np.median(ts))
    # (2) the reference-signature operator: plan + upload + launch + download + fragments + in-place results
    yi = image.Image.create_from_array(scene['y'], normalize=False)
    ts = []
    for _ in range(4):
        objs = [objects.Object() for _ in scene['footprints']]
        for o, fp in zip(objs, scene['footprints']):
            o.footprint = set(fp)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        objects.compute_objects(objs, yi, scene['atoms'], scene['dsm_cfg'], None, out='muted')
        ts.append((time.perf_counter() - t1) * 1e3)
    ex['compute_objects_wall_ms'] = float(np.median(ts[1:]))
    ex['compute_objects_candidates'] = len(scene['footprints'])
    # (3) the stage: all generations, host set cover, downloads -- wall clock per image, alone and 8 images in lock step
    beta, pruning = 150.0, 'isbi24'
    if args.workload in ('bbbc039_like', 'synthetic256'):
        stage = globalenergymin.GlobalEnergyMinimization()
        cfg = config.Config({'global-energy-minimization': {'beta': beta, 'pruning': pruning}})
        mk = lambda: dict(y=scene['y'], y_mask=np.ones(scene['y'].shape, bool), atoms=scene['atoms'], adjacencies=scene['adjacencies'], dsm_cfg=scene['dsm_cfg'])
        stage(mk(), cfg, out='muted')
        ts = []
        for _ in range(3):
            d = mk()
            t1 = time.perf_counter()
            stage(d, cfg, out='muted')
            ts.append((time.perf_counter() - t1) * 1e3)
        ex['stage_wall_ms_per_image'] = float(np.median(ts))
        ex['stage_engine_batches'] = int(getattr(d['performance'], 'engine_batches', 0))
        ex['stage_candidates_solved_ahead_in_vain'] = int(getattr(d['performance'], 'speculative_object_count', 0))
        cfg0 = config.Config({'global-energy-minimization': {'beta': beta, 'pruning': pruning, 'speculation': 0}})
        ts = []
        for _ in range(3):
            d = mk()
            t1 = time.perf_counter()
            stage(d, cfg0, out='muted')
            ts.append((time.perf_counter() - t1) * 1e3)
        ex['stage_wall_ms_per_image_reference_batches'] = float(np.median(ts))       # speculation: 0 = one batch per generation, as the reference
        ts = []
        for _ in range(3):                           # (median of three: a collection of the Python heap in the middle of the image threads costs milliseconds)
            ds = [mk() for _ in range(n_images)]
            gc.collect()
            t1 = time.perf_counter()
            stage.process_many(ds, cfg, out='muted')
            ts.append((time.perf_counter() - t1) * 1e3 / n_images)
        ex[f'stage_wall_ms_per_image_lockstep{n_images}'] = float(np.median(ts))
        if scenes is not None and len({id(sc) for sc in scenes}) > 1:       # the same on the DIFFERENT images of the headline step
            mks = lambda sc: dict(y=sc['y'], y_mask=np.ones(sc['y'].shape, bool), atoms=sc['atoms'], adjacencies=sc['adjacencies'], dsm_cfg=sc['dsm_cfg'])
            ts, ncand = [], 0
            for _ in range(3):
                ds = [mks(sc) for sc in scenes]
                gc.collect()
                t1 = time.perf_counter()
                stage.process_many(ds, cfg, out='muted')
                ts.append((time.perf_counter() - t1) * 1e3 / len(scenes))
                ncand = sum(int(d['performance'].overall_computed_object_count) for d in ds)
            ex[f'stage_wall_ms_per_image_lockstep{len(scenes)}_different_images'] = float(np.median(ts))
            ex['stage_candidates_of_the_different_images'] = ncand
        ex['stage_pruning'] = pruning
        ex['stage_beta'] = beta
    # (3b) the step of rounds 1-2 for comparison: ONE launch over 8 copies of this image (identical work per image, no large clusters)
    if args.workload == 'bbbc039_like':
        L = _capi_lib()
        imgs8 = [engine.DeviceImage(scene['y'], None, scene['atoms'], scene['dsm_cfg']['background_margin']) for _ in range(8)]
        b8 = engine.Batch(imgs8, scene['footprints'] * 8, scene['dsm_cfg'], image_of=np.repeat(np.arange(8, dtype=np.int32), len(scene['footprints'])))
        for _ in range(3):
            b8.launch()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            b8.launch()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) / 20 * 1e3
        L.sdsm_enable_kernel_timing(1)
        km = []
        for _ in range(3):
            b8.launch()
            km.append(L.sdsm_last_solve_kernel_ms())
        L.sdsm_enable_kernel_timing(0)
        r8 = b8.records()
        alg8 = engine.algorithmic_bytes(r8, b8.mask_info)
        ex['step_of_rounds_1_2_eight_copies_of_one_image'] = dict(candidates=len(r8), ms_per_step=ms, candidate_solves_per_s=len(r8) / (ms * 1e-3), solve_kernels_ms=float(np.mean(km)),
                                                                  roofline_frac=alg8 / (float(np.mean(km)) * 1e-3) / 8e12, algorithmic_bytes_per_launch=alg8,
                                                                  note='round 2 reported 639 k solves/s, 5.06 ms solve kernel, 5.8 % on this step (the headline step of rounds 1-2)')
    # (4) preprocessing: 16 B / pixel algorithmic (read g, write y)
    rng = np.random.default_rng(0)
    pre = {}
    for shape, sigma2 in (((520, 696), 10.0), ((4096, 4096), 10.0)):
        g = torch.as_tensor(rng.random(shape)).cuda()
        for _ in range(2):
            engine.preprocess(g, sigma2=sigma2, return_tensor=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            engine.preprocess(g, sigma2=sigma2, return_tensor=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) / reps * 1e3
        px = shape[0] * shape[1]
        # which bound: the separable filters are SciPy's sums bit for bit (unfused multiply and add, centre + R symmetric pairs per axis): 3 R + 1 FP64
        # vector instructions per pixel, axis and filter -- three filters (sigma2 twice, sigma1 once).  The FP64 vector pipes issue 78.6 / 2 = 39.3 T
        # of those per second (the peak counts a fused multiply-add as two); that bound is far above the 16 B / pixel HBM figure
        r1, r2 = int(4 * np.sqrt(2) + 0.5), int(4 * sigma2 + 0.5)
        ops = 2 * 2 * (3 * r2 + 1) + 2 * (3 * r1 + 1)
        pre[f'{shape[0]}x{shape[1]}'] = dict(ms=ms, sigma2=sigma2, algorithmic_GBps=16 * px / (ms * 1e-3) / 1e9, frac_of_8TBps=16 * px / (ms * 1e-3) / 8e12,
                                             fp64_vector_instructions_per_pixel=ops, frac_of_fp64_vector_issue_39_3T=ops * px / (ms * 1e-3) / 39.3e12,
                                             bound='fp64 vector issue (unfused SciPy-order sums), not HBM')
    ex['preprocess'] = pre
    # (4) the deployment DESIGN section 6 recommends for the stage (it is host bound per process): R worker processes share the card, 8 different
    # BBBC039-like images each, one gather of the results -- measured by FRESH child processes of this run (`--mode image_set --ranks-per-gpu R`)
    if args.workload == 'bbbc039_like' and not args.no_ranks_per_gpu:
        for R in (1, 2, 4):
            try:
                res = subprocess.run([sys.executable, os.path.abspath(__file__), '--mode', 'image_set', '--workload', 'bbbc039_like', '--images-workload-explicit',
                                      '--images', '8', '--gpus', '1', '--ranks-per-gpu', str(R), '--repeats', '3'], capture_output=True, timeout=240,
                                     env={k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')})
                line = [l for l in res.stdout.decode().splitlines() if l.startswith('{')][-1]
                d = json.loads(line)
                ex[f'stage_wall_ms_per_image_ranks_per_gpu_{R}'] = d['config']['wall_ms_per_image']
                ex[f'stage_candidate_solves_per_s_ranks_per_gpu_{R}'] = d['value']
            except Exception as e:                              # noqa: BLE001 -- an extra, not the headline
                ex[f'stage_wall_ms_per_image_ranks_per_gpu_{R}'] = f'failed: {type(e).__name__}'
    return ex


def launch_figures(scene, n_timed=15, copies=1):
    """One engine launch over all candidates of one image -- or of `copies` images of that shape in one plan -- (throughput scheduling, as
    the headline step): wall clock per launch, HIP-event time of the solve kernels, solves/s and the roofline fraction by SURVEY.md 8(d)'s
    byte count."""
    import torch
    from superdsm_amd import _capi, engine
    L = _capi.lib()
    img = engine.DeviceImage(scene['y'], None, scene['atoms'], scene['dsm_cfg']['background_margin'])
    if copies > 1:
        batch = engine.Batch([img] * copies, scene['footprints'] * copies, scene['dsm_cfg'], image_of=np.repeat(np.arange(copies, dtype=np.int32), len(scene['footprints'])))
    else:
        batch = engine.Batch(img, scene['footprints'], scene['dsm_cfg'])
    for _ in range(4):
        batch.launch()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n_timed):
        t1 = time.perf_counter()
        batch.launch()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t1) * 1e3)
    L.sdsm_enable_kernel_timing(1)
    km, sm = [], []
    for _ in range(3):
        batch.launch()
        km.append(L.sdsm_last_solve_kernel_ms())
        sm.append(L.sdsm_last_setup_kernel_ms())
    L.sdsm_enable_kernel_timing(0)
    recs = batch.records()
    alg = engine.algorithmic_bytes(recs, batch.mask_info)
    ms = float(np.median(ts))
    kern = float(np.mean(km))
    return dict(image=f'{scene["y"].shape[0]}x{scene["y"].shape[1]}', images_per_launch=copies, candidates=len(scene['footprints']) * copies, ms_per_launch=ms, solve_kernels_ms=kern, setup_kernel_ms=float(np.mean(sm)),
                candidate_solves_per_s=len(scene['footprints']) * copies / (ms * 1e-3), median_N=int(np.median(recs['n_pixels'])), max_N=int(recs['n_pixels'].max()),
                median_M=int(np.median(recs['n_deform'])), max_M=int(recs['n_deform'].max()),
                roofline={'bound': 'hbm', 'achieved': alg / (kern * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s', 'frac': alg / (kern * 1e-3) / 8e12, 'algorithmic_bytes_per_launch': alg},
                status_counts={str(k): int(v) for k, v in zip(*np.unique(recs['status'], return_counts=True))})


def other_configs(args):
    """The BASELINE.json configs beside the headline one, each measured in this same run."""
    import torch
    from superdsm_amd import config, globalenergymin, testing
    out = {}
    sc = testing.make_scene('gowt1_like', max_size=3)
    out['gowt1_like'] = dict(launch_figures(sc), stands_for='BASELINE.json configs[2]: GOWT1-1 frame (sparse nuclei, large regions), one launch = all candidates of the frame')
    sc = testing.make_scene('nih3t3_like', max_size=2)
    fig = launch_figures(testing.make_scene('nih3t3_like', max_size=3))
    stage = globalenergymin.GlobalEnergyMinimization()
    cfg = config.Config({'global-energy-minimization': {'beta': 1200.0, 'pruning': 'isbi24'}})
    mk = lambda i: dict(y=np.ascontiguousarray(sc['y'] * (1 - 0.003 * i)), y_mask=np.ones(sc['y'].shape, bool), atoms=sc['atoms'], adjacencies=sc['adjacencies'], dsm_cfg=sc['dsm_cfg'])
    stage.process_many([mk(0)], cfg, out='muted')
    ts, ncand = [], 0
    for _ in range(3):
        ds = [mk(i) for i in range(3)]
        gc.collect()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        stage.process_many(ds, cfg, out='muted')
        ts.append((time.perf_counter() - t1) * 1e3 / 3)
        ncand = sum(int(d['performance'].overall_computed_object_count) for d in ds)
    out['nih3t3_like'] = dict(fig, stands_for='BASELINE.json configs[3] on ONE GPU: NIH3T3-like images; launch figures for all candidates of one image, and the whole '
                                              'global-energy-minimisation stage on a set of 3 images in lock step (image sets over N GPUs: --mode image_set)',
                              stage_wall_ms_per_image_set_of_3=float(np.median(ts)), stage_candidates_set_of_3=ncand)
    sc = testing.make_scene('synthetic512', max_size=3)
    out['synthetic512'] = dict(launch_figures(sc, copies=16), single_image=launch_figures(sc),
                               stands_for="BASELINE.json north_star: 'candidate-solves/sec on synthetic 512x512 nuclei images': 60 nuclei of radius 15 (some overlapping), "
                                          'every candidate; one launch over 16 such images (a plan holds up to 16), and one image alone')
    sc = testing.make_scene('synthetic4096', max_size=3)
    out['synthetic4096'] = dict(launch_figures(sc, n_timed=3), stands_for='BASELINE.json configs[4] on ONE GPU: synthetic 4096x4096, ~2000 dense overlapping nuclei, every candidate in one launch '
                                                                           '(over N GPUs as ONE sharded batch: --mode sharded)')
    return out


def sharded_mode(args, world, rank, backend, n_gpus_used):
    """BASELINE.json configs[4] as specified (strong scaling): the candidates of ONE batch dealt to the ranks by cost (dist.Sharder),
    every rank solves its share on its GPU, ONE all-gather of records + bit-packed masks per step.  value = candidates of the batch / s."""
    import torch
    import torch.distributed as dist
    from superdsm_amd import engine, testing
    from superdsm_amd import dist as sdist
    wl = args.workload if args.workload != 'bbbc039_like' or args.images_workload_explicit else 'synthetic4096'
    scene = testing.make_scene(wl, max_size=args.max_size)
    img = engine.DeviceImage(scene['y'], None, scene['atoms'], scene['dsm_cfg']['background_margin'])
    if world == 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        dist.init_process_group(backend, rank=0, world_size=1)
    sb = sdist.Sharder().prepare(img, scene['footprints'], scene['dsm_cfg'])
    for _ in range(max(1, args.warmup)):
        sb.step()
    torch.cuda.synchronize()
    region_ms = []
    for _ in range(max(1, min(args.repeats, 3))):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sb.step()
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        region_ms.append(float(tmax.item()) * 1e3)
    recs, _ = sb.results()
    if rank == 0:
        dt = float(np.median(region_ms)) * 1e-3
        n = len(scene['footprints'])
        alg = engine.algorithmic_bytes(recs, sb.mask_info)
        print(json.dumps({
            'metric': 'candidate DSM solves/sec', 'value': n * args.steps / dt, 'unit': 'candidate solves/s', 'n_gpus': n_gpus_used, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'{wl} {scene["y"].shape[0]}x{scene["y"].shape[1]}: ONE batch of {n} candidates dealt to {world} ranks by cost (snake order), '
                                   f'one all-gather of records + masks per step ({backend})', 'candidates': n, 'ranks': world, 'ranks_per_gpu': max(1, args.ranks_per_gpu),
                       'shard_sizes': [len(sh) for sh in sb.shards], 'payload_bytes_per_rank': int(sb.pad), 'backend': backend,
                       'timed_regions_ms': region_ms},
            'roofline': {'bound': 'hbm', 'achieved': alg / (dt / args.steps) / 1e9 / max(1, n_gpus_used), 'peak': 8000.0, 'unit': 'GB/s per GPU (whole step incl. setup kernel and the gather)',
                         'frac': alg / (dt / args.steps) / 8e12 / max(1, n_gpus_used), 'traffic': None, 'algorithmic_bytes_per_step': alg},
            'status_counts': {str(k): int(v) for k, v in zip(*np.unique(recs['status'], return_counts=True))}, 'cpu_baseline': None}))
    dist.destroy_process_group()


def image_set_mode(args, world, rank, backend, n_gpus_used):
    """BASELINE.json configs[3]: a set of NIH3T3-like images dealt to the ranks; every rank runs the stage on its images in lock step,
    ONE gather of the per-image results (cover footprints, energies) at the end.  value = candidate solves / s over the whole set."""
    import torch
    import torch.distributed as dist
    from superdsm_amd import config, globalenergymin, testing
    from superdsm_amd import dist as sdist
    wl = args.workload if args.workload != 'bbbc039_like' or args.images_workload_explicit else 'nih3t3_like'
    scene = testing.make_scene(wl, max_size=2 if wl != 'bbbc039_like' else 3)
    # BBBC039-like sets: DIFFERENT images (the eight layouts of the headline step), else copies of one scene with scaled intensities
    scenes = [testing.make_scene(wl, max_size=3, layout_index=k) for k in range(8)] if wl == 'bbbc039_like' else [scene]
    n_images = (args.images or 4) * world
    mine = sdist.deal_images(n_images, world)[rank]
    beta = {'nih3t3_like': 1200.0, 'gowt1_like': 1188.0}.get(wl, 150.0)
    stage = globalenergymin.GlobalEnergyMinimization()
    cfg = config.Config({'global-energy-minimization': {'beta': beta, 'pruning': 'isbi24'}})
    def mk(i):
        sc = scenes[i % len(scenes)]
        return dict(y=np.ascontiguousarray(sc['y'] * (1 - 0.003 * (i % 7))) if len(scenes) == 1 else sc['y'], y_mask=np.ones(sc['y'].shape, bool), atoms=sc['atoms'],
                    adjacencies=sc['adjacencies'], dsm_cfg=sc['dsm_cfg'])
    stage.process_many([mk(i) for i in mine[:1]], cfg, out='muted')            # warm-up: one image
    torch.cuda.synchronize()
    times = []
    for _ in range(max(1, min(args.repeats, 3))):
        datas = [mk(i) for i in mine]
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stage.process_many(datas, cfg, out='muted')
        local = [(i, sorted(sorted(int(a) for a in o.footprint) for o in d['cover'].solution), float(d['cover'].costs),
                  int(d['performance'].overall_computed_object_count)) for i, d in zip(mine, datas)]
        got = sdist.gather_objects(local, dst=0) if world > 1 else [local]   # the one gather of the run
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        times.append(dt)
    if rank == 0:
        dt = float(np.median(times))
        flat = sorted(x for part in got for x in part)
        ncand = sum(x[3] for x in flat)
        print(json.dumps({
            'metric': 'candidate DSM solves/sec', 'value': ncand / dt, 'unit': 'candidate solves/s', 'n_gpus': n_gpus_used, 'steps': 1, 'warmup': 1,
            'ms_per_step': dt * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'{wl} image set (BASELINE.json configs[3] stand-in): {n_images} images of {scene["y"].shape[0]}x{scene["y"].shape[1]}, '
                                   f'{n_images // world} per rank, global-energy-minimisation stage in lock step, one gather of the results at the end',
                       'images': n_images, 'candidates_solved': ncand, 'ranks': world, 'ranks_per_gpu': max(1, args.ranks_per_gpu), 'backend': backend if world > 1 else None, 'wall_ms_per_image_and_rank': dt * 1e3 / (n_images // world),
                       'wall_ms_per_image': dt * 1e3 / n_images * n_gpus_used, 'objects_in_covers': sum(len(x[1]) for x in flat)},
            'roofline': None, 'cpu_baseline': None}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
