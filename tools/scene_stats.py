#!/usr/bin/env python3
"""Solve a workload's candidates on the GPU, print size / status / timing statistics, and spot-check a sample against
the CPU oracle.  usage: python tools/scene_stats.py gowt1_like [n_oracle_samples]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superdsm_amd import testing
from oracle import oracle
wl = sys.argv[1]; ns = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = time.time(); scene = testing.make_scene(wl, max_size=3); print(wl, 'scene', round(time.time() - t0, 1), 's', scene['y'].shape, 'atoms', scene['atoms'].max(), 'cands', len(scene['footprints']), scene['dsm_cfg'])
res = testing.solve_scene_gpu(scene)
torch.cuda.synchronize(); t0 = time.time(); res['batch'].launch(); torch.cuda.synchronize(); print('GPU batch ms', round((time.time() - t0) * 1e3, 2))
r = res['records']
print('status counts', dict(zip(*np.unique(r['status'], return_counts=True))), 'N med/max', int(np.median(r['n_pixels'])), r['n_pixels'].max(), 'M med/max', int(np.median(r['n_deform'])), r['n_deform'].max(),
      'iters_dsm med/max', int(np.median(r['iters_dsm'])), r['iters_dsm'].max())
idx = np.linspace(0, len(scene['footprints']) - 1, ns).astype(int)
t0 = time.time()
orecs, ofr, _ = oracle.compute_objects(scene['y'], None, scene['atoms'], [scene['footprints'][i] for i in idx], scene['dsm_cfg'], nthreads=0)
print('oracle', round(time.time() - t0, 1), 's for', ns)
for j, k in enumerate(idx):
    tol = 1e-6 * orecs['N'][j] / 1000 + 1e-5 * abs(orecs['energy'][j])
    dice = testing.dice(res['fragments'][k][0], res['fragments'][k][1], orecs['fg_offset'][j], ofr[j], scene['y'].shape)
    print(k, 'N', r['n_pixels'][k], orecs['N'][j], 'M', r['n_deform'][k], orecs['M'][j], 'E', r['energy'][k], orecs['energy'][j], 'ok' if abs(r['energy'][k] - orecs['energy'][j]) <= tol else 'DIFF', 'dice %.5f' % dice, 'st', r['status'][k], orecs['status'][j], 'it', r['iters_dsm'][k], orecs['iters_dsm'][j])
