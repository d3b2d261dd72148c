#!/usr/bin/env python3
"""Where the host time of the product entry points goes: cProfile of objects.compute_objects (501 candidates) and of
GlobalEnergyMinimization on the BBBC039-like scene.  usage: python tools/profile_host.py"""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superdsm_amd import config, globalenergymin, image, objects, testing

scene = testing.make_scene('bbbc039_like', max_size=3)
y = image.Image.create_from_array(scene['y'], normalize=False)


def make():
    objs = []
    for fp in scene['footprints']:
        o = objects.Object()
        o.footprint = set(int(a) for a in fp)
        objs.append(o)
    return objs


def prof(fn, n=25):
    pr = cProfile.Profile()
    pr.enable()
    fn()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(n)
    print(s.getvalue()[:6000])
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(n)
    print(s.getvalue()[:6000])


for rep in range(3):
    objs = make()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    objects.compute_objects(objs, y, scene['atoms'], scene['dsm_cfg'], None, out='muted')
    print(f'compute_objects({len(objs)}): {(time.perf_counter() - t0) * 1e3:.1f} ms')
objs = make()
prof(lambda: objects.compute_objects(objs, y, scene['atoms'], scene['dsm_cfg'], None, out='muted'))
stage = globalenergymin.GlobalEnergyMinimization()
cfg = config.Config({'global-energy-minimization': {'beta': 150.0, 'pruning': 'isbi24'}})
mk = lambda: dict(y=scene['y'], y_mask=np.ones(scene['y'].shape, bool), atoms=scene['atoms'], adjacencies=scene['adjacencies'], dsm_cfg=scene['dsm_cfg'])
stage(mk(), cfg, out='muted')
t0 = time.perf_counter()
d = mk()
stage(d, cfg, out='muted')
print(f'stage: {(time.perf_counter() - t0) * 1e3:.1f} ms, {d["performance"].overall_computed_object_count} candidates')
prof(lambda: stage(mk(), cfg, out='muted'), 40)
