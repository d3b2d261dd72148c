"""Diagnostic: wall clock of the global-energy-minimisation stage, one BBBC039-like image alone and 8 different images in lock step, for the settings given
as environment assignments on the command line (each measured in this process, alternating):   python tools/stage_ab.py SDSM_LOCKSTEP_TOGGLES=1 SDSM_LOCKSTEP_TOGGLES=0"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from superdsm_amd import config, globalenergymin, testing
settings = [dict(kv.split('=') for kv in a.split(',')) for a in sys.argv[1:]] or [{}]
scenes = [testing.make_scene('bbbc039_like', max_size=3, layout_index=k) for k in range(8)]
stage = globalenergymin.GlobalEnergyMinimization()
cfg = config.Config({'global-energy-minimization': {'beta': 150.0, 'pruning': 'isbi24'}})
mk = lambda sc: dict(y=sc['y'], y_mask=np.ones(sc['y'].shape, bool), atoms=sc['atoms'], adjacencies=sc['adjacencies'], dsm_cfg=sc['dsm_cfg'])
stage(mk(scenes[0]), cfg, out='muted')
stage.process_many([mk(sc) for sc in scenes], cfg, out='muted')
res = {i: ([], []) for i in range(len(settings))}
for rep in range(6):
    for i, st in enumerate(settings):
        os.environ.update(st)
        for _ in range(3):
            d = mk(scenes[0]); torch.cuda.synchronize(); t = time.perf_counter(); stage(d, cfg, out='muted'); res[i][0].append((time.perf_counter() - t) * 1e3)
        ds = [mk(sc) for sc in scenes]; gc.collect(); torch.cuda.synchronize(); t = time.perf_counter(); stage.process_many(ds, cfg, out='muted'); res[i][1].append((time.perf_counter() - t) * 1e3 / 8)
for i, st in enumerate(settings):
    a, b = res[i]
    print(st, 'stage alone: median %.2f min %.2f ms | 8 different images in lock step: median %.2f min %.2f ms per image' % (np.median(a), np.min(a), np.median(b), np.min(b)))
