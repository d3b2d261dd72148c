#!/usr/bin/env python3
"""Diagnostic: the engine batches of one GlobalEnergyMinimization run on the BBBC039-like scene -- candidates, largest region,
time from launch to downloaded results.  usage: python tools/stage_batches.py [workload [beta [generations solved ahead]]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superdsm_amd import config, engine, globalenergymin, testing

wl = sys.argv[1] if len(sys.argv) > 1 else 'bbbc039_like'
scene = testing.make_scene(wl, max_size=3, layout_index=int(os.environ.get('LAYOUT', 0)))      # LAYOUT=k: the k-th BBBC039-like layout
stage = globalenergymin.GlobalEnergyMinimization()
gem = {'beta': float(sys.argv[2]) if len(sys.argv) > 2 else 150.0, 'pruning': 'isbi24'}
if len(sys.argv) > 3:
    gem['speculation'] = int(sys.argv[3])
cfg = config.Config({'global-energy-minimization': gem})
mk = lambda: dict(y=scene['y'], y_mask=np.ones(scene['y'].shape, bool), atoms=scene['atoms'], adjacencies=scene['adjacencies'], dsm_cfg=scene['dsm_cfg'])
stage(mk(), cfg, out='muted')
log = []
orig_launch, orig_download = engine.Batch.launch, engine.Batch.download
def launch(self, *a, **k):
    self._t0 = time.perf_counter()
    return orig_launch(self, *a, **k)
def download(self, *a, **k):
    r = orig_download(self, *a, **k)
    log.append((len(r[0]), int(r[0]['n_pixels'].max()), int(r[0]['n_deform'].max()), (time.perf_counter() - self._t0) * 1e3, float(np.median(r[0]['iters_dsm']))))
    return r
engine.Batch.launch, engine.Batch.download = launch, download
t0 = time.perf_counter()
stage(mk(), cfg, out='muted')
print(f'stage {1e3 * (time.perf_counter() - t0):.1f} ms')
for n, npx, m, ms, it in log:
    print(f'  batch of {n:4d} candidates, largest region {npx:6d} px, largest M {m:3d}, median DSM iterations {it:.0f}: {ms:.2f} ms launch -> results')
print(f'  sum {sum(l[3] for l in log):.1f} ms')
