#!/usr/bin/env python3
"""Diagnostic (no GPU): the host logic of the global-energy-minimisation stage alone -- enumeration, pruning, set cover -- on the
BBBC039-like adjacency graph with a table of pseudo-energies instead of the solver.  usage: python tools/host_logic_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, hashlib, cProfile, pstats, io
from superdsm_amd import globalenergymin as gem, testing
scene = testing.make_scene('bbbc039_like', max_size=3)
adj = scene['adjacencies']
def h(s): return int(hashlib.sha1(s.encode()).hexdigest()[:8], 16) / 2**32
cache = {}
def fake_energy(fp):
    k = frozenset(fp)
    if k not in cache:
        f = sorted(int(a) for a in fp)
        cache[k] = sum(30 + 20 * h(f'a{a}') for a in f) * (0.75 + 0.5 * h(','.join(map(str, f)))) if len(f) > 1 else 30 + 20 * h(f'a{f[0]}')
    return cache[k]
frag = (np.zeros(2, int), np.zeros((1, 1), bool))
def fake(objs, y, atoms_map, dsm_cfg, log_root_dir, status_line=None, out=None, shard=None):
    for o in objs:
        o.energy = fake_energy(o.footprint); o.is_optimal, o.on_boundary, o.processing_time = True, False, 0
        o.fg_offset, o.fg_fragment = frag
gem.compute_objects = fake
for spec in (0, 2):
    dts=[]
    for rep in range(30):
        t0 = time.perf_counter()
        gens, costs, cover, objs, perf = gem._compute_generations(adj, None, scene['atoms'], None, 'isbi24', {}, beta=5.0, out='muted', speculation=spec)
        dts.append((time.perf_counter() - t0) * 1e3)
    dt=min(dts)
    print(f'speculation {spec}: host logic {dt:.2f} ms, {len(objs)} objects, {len(gens)} generations')
pr = cProfile.Profile(); pr.enable()
gem._compute_generations(adj, None, scene['atoms'], None, 'isbi24', {}, beta=5.0, out='muted', speculation=0)
pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:2500])
