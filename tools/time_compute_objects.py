#!/usr/bin/env python3
"""Wall clock of superdsm_amd.objects.compute_objects (the reference-signature entry point) for all candidates of a
workload, split into its host and device parts.  usage: python tools/time_compute_objects.py [workload]"""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superdsm_amd import image, objects, testing

wl = sys.argv[1] if len(sys.argv) > 1 else 'bbbc039_like'
scene = testing.make_scene(wl, max_size=3)
y = image.Image.create_from_array(scene['y'], normalize=False)
cfg = dict(scene['dsm_cfg'], cachesize=1, cp_timeout=300, smooth_mat_max_allocations=np.inf)


def make():
    objs = []
    for fp in scene['footprints']:
        o = objects.Object()
        o.footprint = set(int(a) for a in fp)
        objs.append(o)
    return objs


for rep in range(3):
    objs = make()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    objects.compute_objects(objs, y, scene['atoms'], cfg, None, out='muted')
    torch.cuda.synchronize()
    print(f'compute_objects({len(objs)} candidates): {(time.perf_counter() - t0) * 1e3:.1f} ms wall')
pr = cProfile.Profile()
objs = make()
pr.enable()
objects.compute_objects(objs, y, scene['atoms'], cfg, None, out='muted')
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18)
print(s.getvalue()[:3500])
