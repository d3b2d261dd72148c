#!/usr/bin/env python3
"""Diagnostic: GlobalEnergyMinimization.process_many on n copies of the BBBC039-like image -- wall clock per image, multi-image
batches, candidates per batch -- for several settings of `speculation`.  usage: python tools/lockstep_probe.py [n_images]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superdsm_amd import config, globalenergymin, objects, testing

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scene = testing.make_scene('bbbc039_like', max_size=3)
stage = globalenergymin.GlobalEnergyMinimization()
mk = lambda: dict(y=scene['y'], y_mask=np.ones(scene['y'].shape, bool), atoms=scene['atoms'], adjacencies=scene['adjacencies'], dsm_cfg=scene['dsm_cfg'])
sizes = []
orig = objects.compute_objects_multi
def spy(jobs, *a, **k):
    t0 = time.perf_counter()
    r = orig(jobs, *a, **k)
    sizes.append((len(jobs), sum(len(j[0]) for j in jobs), (time.perf_counter() - t0) * 1e3))
    return r
globalenergymin.compute_objects_multi = spy
for spec in (0, 1, None, 0, 1):
    gem = {'beta': 150.0, 'pruning': 'isbi24'}
    if spec is not None:
        gem['speculation'] = spec
    cfg = config.Config({'global-energy-minimization': gem})
    sizes.clear()
    ds = [mk() for _ in range(n)]
    t0 = time.perf_counter()
    stage.process_many(ds, cfg, out='muted')
    dt = (time.perf_counter() - t0) * 1e3
    print(f'speculation={spec}: {dt / n:.1f} ms per image, {len(sizes)} multi-image batches, GPU wait {sum(s[2] for s in sizes):.1f} ms: ' + ' '.join(f'{a}x{b}({c:.1f})' for a, b, c in sizes))
