"""Diagnostic: wall time of the stage on single BBBC039-like images against the depth / budget of the generations solved ahead (PRUNING=exact|isbi24)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superdsm_amd import globalenergymin, testing
from superdsm_amd.output import get_output
out = get_output('muted')
stage = globalenergymin.GlobalEnergyMinimization()
for layout in (0, 2, 5):
    s = testing.make_scene('bbbc039_like', max_size=3, layout_index=layout)
    mk = lambda: dict(y=s['y'], y_mask=np.ones(s['y'].shape, bool), atoms=s['atoms'], adjacencies=s['adjacencies'], dsm_cfg=s['dsm_cfg'])
    for depth, budget in ((8, 2048),):
        cfg = {'beta': 150.0, 'pruning': os.environ.get('PRUNING', 'isbi24'), 'speculation': depth}
        for _ in range(3): r = stage.process(mk(), cfg, out, None, speculation_budget=budget)
        ts = []
        for _ in range(10):
            d = mk(); t0 = time.perf_counter(); r = stage.process(d, cfg, out, None, speculation_budget=budget); ts.append((time.perf_counter() - t0) * 1e3)
        p = r['performance']
        nb, nv = getattr(p, 'engine_batches', '-'), getattr(p, 'speculative_object_count', '-')
        print(f'layout {layout} depth {depth} budget {budget}: median {np.median(ts):.2f} ms, batches {nb}, in vain {nv}, computed {p.overall_computed_object_count}', flush=True)
