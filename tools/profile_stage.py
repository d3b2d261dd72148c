#!/usr/bin/env python3
"""Diagnostic: where the wall clock of ONE image's GlobalEnergyMinimization.process goes: inside compute_objects (engine batch:
plan, upload, launch, wait, records, objects) against the generation logic around it; cProfile of both.
usage: python tools/profile_stage.py [layout]"""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superdsm_amd import config, globalenergymin, objects, testing

k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
s = testing.make_scene('bbbc039_like', max_size=3, layout_index=k)
stage = globalenergymin.GlobalEnergyMinimization()
mk = lambda: dict(y=s['y'], y_mask=np.ones(s['y'].shape, bool), atoms=s['atoms'], adjacencies=s['adjacencies'], dsm_cfg=s['dsm_cfg'])
cfg = {'beta': 150.0, 'pruning': 'isbi24'}
calls = []
orig = objects.compute_objects
inner = cProfile.Profile()
def spy(objs, *a, **kw):
    t0 = time.perf_counter()
    inner.enable()
    r = orig(objs, *a, **kw)
    inner.disable()
    calls.append((len(list(objs)) if not hasattr(objs, '__len__') else len(objs), (time.perf_counter() - t0) * 1e3))
    return r
spy.__module__ = orig.__module__
from superdsm_amd.output import get_output
out = get_output('muted')
for _ in range(3):
    stage.process(mk(), cfg, out, None)
ts = []
for _ in range(10):
    d = mk(); t0 = time.perf_counter(); stage.process(d, cfg, out, None); ts.append((time.perf_counter() - t0) * 1e3)
print('stage wall per image: median %.2f ms (min %.2f)' % (np.median(ts), min(ts)))
globalenergymin.compute_objects = spy
outer = cProfile.Profile()
d = mk(); t0 = time.perf_counter(); outer.enable(); stage.process(d, cfg, out, None); outer.disable(); dt = (time.perf_counter() - t0) * 1e3
print('profiled run: %.2f ms; compute_objects calls (candidates, ms):' % dt, calls, 'sum %.2f ms' % sum(c[1] for c in calls))
for name, pr in (('whole stage', outer), ('inside compute_objects', inner)):
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('tottime').print_stats(22); print('----', name); print(st.getvalue()[:5000])
