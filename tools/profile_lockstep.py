#!/usr/bin/env python3
"""Diagnostic: where the wall clock of GlobalEnergyMinimization.process_many (8 BBBC039-like images in lock step) goes: time inside
compute_objects_multi (plan, launch, wait, fragments, results) against the rest (host logic of the image threads), and a cProfile of
compute_objects_multi itself.  usage: python tools/profile_lockstep.py [n_images]"""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superdsm_amd import config, globalenergymin, objects, testing

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scenes = [testing.make_scene('bbbc039_like', max_size=3, layout_index=k % 8) for k in range(n)]
stage = globalenergymin.GlobalEnergyMinimization()
mk = lambda s: dict(y=s['y'], y_mask=np.ones(s['y'].shape, bool), atoms=s['atoms'], adjacencies=s['adjacencies'], dsm_cfg=s['dsm_cfg'])
cfg = config.Config({'global-energy-minimization': {'beta': 150.0, 'pruning': 'isbi24'}})
calls = []
orig = objects.compute_objects_multi
pr = cProfile.Profile()
def spy(jobs, *a, **k):
    t0 = time.perf_counter()
    pr.enable()
    r = orig(jobs, *a, **k)
    pr.disable()
    calls.append((len(jobs), sum(len(j[0]) for j in jobs), (time.perf_counter() - t0) * 1e3))
    return r
stage.process_many([mk(s) for s in scenes], cfg, out='muted')        # warm-up
for rep in range(3):
    calls.clear()
    ds = [mk(s) for s in scenes]
    globalenergymin.compute_objects_multi = spy if rep == 2 else orig
    t0 = time.perf_counter()
    stage.process_many(ds, cfg, out='muted')
    dt = (time.perf_counter() - t0) * 1e3
    print(f'run {rep}: {dt:.1f} ms for {n} images = {dt / n:.2f} ms per image')
print('multi-image batches (images x candidates (ms inside compute_objects_multi)):', ' '.join(f'{a}x{b}({c:.1f})' for a, b, c in calls), ' sum %.1f ms' % sum(c[2] for c in calls))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
print(s.getvalue()[:4500])
