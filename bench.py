#!/usr/bin/env python3
"""Headline benchmark: candidate DSM solves per second on the BBBC039-like 520x696 image (BASELINE.json
configs[1]), all candidates of the image in one batch per step, 1..N MI355X (one process per GPU).

A "step" = one pass of the hot path over the image's whole candidate list with the image already resident in
HBM: region crops, greedy grids + G~ rows, elliptical + DSM solves, masks, records (one sdsm_batch_launch),
plus -- for N > 1 -- the single gather of the fixed-size records and bit-packed masks to rank 0 over RCCL.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Independent steps are queued on separate HIP streams; ROCm maps streams onto 4 hardware queues by default and
# kernels of streams that share a queue serialise.  Must be set before the HIP runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=48)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--workload', default='bbbc039_like')
    ap.add_argument('--max-size', type=int, default=3, help='candidates = connected atom subsets up to this size + universes')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='budget of the CPU baseline sample')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--inflight', type=int, default=8, help='independent steps (images) in flight on separate streams; 1 = strictly sequential steps')
    return ap.parse_args()


def cpu_baseline(scene, budget_s):
    """Oracle (CPU restatement, kind "port") on the same candidate list, one OpenMP thread per candidate on every
    host core this process may use; the list is repeated until about `budget_s` seconds of wall time are spent."""
    from oracle import oracle
    fps = scene['footprints']
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    t0 = time.time()
    oracle.compute_objects(scene['y'], None, scene['atoms'], fps, scene['dsm_cfg'], nthreads=cores)
    t_once = time.time() - t0
    reps = int(max(1, min(200, round(budget_s / max(t_once, 1e-3)))))
    t0 = time.time()
    for _ in range(reps):
        oracle.compute_objects(scene['y'], None, scene['atoms'], fps, scene['dsm_cfg'], nthreads=cores)
    dt = time.time() - t0
    return dict(value=reps * len(fps) / dt, unit='candidate solves/s', cores=cores, kind='port',
                sample=f'all {len(fps)} candidates of the same image x {reps} passes, one OpenMP thread per candidate, {dt:.1f} s wall')


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the DSM solve path has no CPU fallback)'
    # rehearsal on a box with fewer GPUs than ranks: SDSM_BENCH_BACKEND=gloo shares the visible devices round-robin
    backend = os.environ.get('SDSM_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend, rank=rank, world_size=world)

    from superdsm_amd import _capi, engine, testing
    from superdsm_amd import dist as sdist

    scene = testing.make_scene(args.workload, max_size=args.max_size)
    fps = scene['footprints']
    img = engine.DeviceImage(scene['y'], None, scene['atoms'], scene['dsm_cfg']['background_margin'])
    # Steps are independent (in production: different images); up to `inflight` of them are queued on separate streams,
    # each with its own workspace / record / mask buffers, so that the GPU is not idle while one image's slowest
    # candidate finishes.  Every step still is one full pass of the hot path over the whole candidate list.
    nfl = max(1, min(args.inflight, args.steps))
    batches = [engine.Batch(img, fps, scene['dsm_cfg']) for _ in range(nfl)]
    streams = [torch.cuda.Stream() for _ in range(nfl)]
    gathers = [sdist.RecordGather(b, world, rank) if world > 1 else None for b in batches]
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

    for k in range(nfl):                       # untimed: touch every in-flight slot once (first launch of a plan: kernel attributes,
        step(k)                                # lazy allocations), so that --warmup smaller than --inflight does not time cold slots
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
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
    # latency of ONE step with nothing else in flight (wall clock per image), scheduled for latency: the largest regions
    # get a 512-thread workgroup each (sdsm_plan_set_latency_mode; same results, fewer solves per second under load)
    lat = engine.Batch(img, fps, scene['dsm_cfg'], latency_mode=True)
    lat.launch()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    lat.launch()
    torch.cuda.synchronize()
    single_ms = (time.perf_counter() - t1) * 1e3
    del lat
    L.sdsm_enable_kernel_timing(1)
    solve_ms = []
    # kernel-level timing with HIP events on the launch stream: a few extra, separately timed launches
    for _ in range(min(5, args.steps)):
        batch.launch()
        solve_ms.append(L.sdsm_last_solve_kernel_ms())
    setup_ms = L.sdsm_last_setup_kernel_ms()
    L.sdsm_enable_kernel_timing(0)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

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
    traffic = None
    pmc_path = os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json')
    if os.path.exists(pmc_path) and args.workload == 'bbbc039_like':
        try:
            traffic = json.load(open(pmc_path)).get('solve_hbm_bytes_per_launch')
        except Exception:
            traffic = None
    out = {
        'metric': 'candidate DSM solves/sec', 'value': value, 'unit': 'candidate solves/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'{args.workload} {scene["y"].shape[0]}x{scene["y"].shape[1]} (BASELINE.json configs[1] stand-in: ellipses at the centres/areas of a '
                               'reference BBBC039 regression CSV), all candidates of the image per step: connected atom subsets of size <= '
                               f'{args.max_size} + cluster universes', 'candidates_per_step_per_gpu': len(fps), 'atoms': int(scene['atoms'].max()),
                   'median_N': int(np.median(recs['n_pixels'])), 'median_M': int(np.median(recs['n_deform'])),
                   'parallelism': f'{world} x (1 process per GPU), candidates sharded by image replica, one RCCL gather per step' if world > 1 else 'single GPU',
                   'steps_in_flight': nfl, 'wall_ms_per_image': single_ms},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s', 'frac': achieved / 8000.0, 'traffic': traffic,
                     'kernel': 'sdsm_k_solve (all three size classes of one launch)', 'kernel_ms': kern_ms, 'setup_kernel_ms': setup_ms,
                     'algorithmic_bytes_per_launch': alg_bytes,
                     'achieved_with_steps_in_flight': alg_bytes * world / (dt / args.steps) / 1e9 / world,   # same bytes over the time per step of the timed region (per GPU)
                     'fp64_vector_tflops': flops / (kern_ms * 1e-3) / 1e12, 'fp64_vector_frac_of_78.6': flops / (kern_ms * 1e-3) / 1e12 / 78.6,
                     'pixel_evaluations_per_launch': int((evals * recs['n_pixels']).sum())},
        'status_counts': {str(k): int(v) for k, v in zip(*np.unique(recs['status'], return_counts=True))},
    }
    if rank == 0 and not args.no_cpu and world == 1:
        out['cpu_baseline'] = cpu_baseline(scene, args.cpu_seconds)
    elif rank == 0:
        out['cpu_baseline'] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
