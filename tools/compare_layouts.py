"""Diagnostic: the eight BBBC039-like layouts, every candidate: GPU records against the CPU oracle (energies, status, passes over the pixels)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superdsm_amd import testing
from oracle import oracle
tot_g = tot_o = 0
for k in range(8):
    sc = testing.make_scene('bbbc039_like', layout_index=k)
    res = testing.solve_scene_gpu(sc)
    recs = res['records']
    orecs, ofr, _ = oracle.compute_objects(sc['y'], None, sc['atoms'], sc['footprints'], sc['dsm_cfg'], nthreads=0)
    ev = recs['evals_full'] + recs['evals_value']
    tol = 1e-6 * orecs['N'] / 1000 + 1e-5 * np.abs(orecs['energy'])
    de = np.abs(recs['energy'] - orecs['energy'])
    bad = np.flatnonzero((de > tol) & ~((orecs['energy'] < 1e-3) & (recs['energy'] <= orecs['energy'] + tol)))
    st = np.flatnonzero((recs['status'] != orecs['status']))
    tot_g += ev.sum(); tot_o += orecs['evals'].sum()
    print(f'layout {k}: {len(recs)} candidates, evals GPU {ev.sum()} oracle {orecs["evals"].sum()}, energy outside tolerance {len(bad)}, status differs {len(st)}, max evals GPU {ev.max()} oracle {orecs["evals"].max()}')
    for i in np.argsort(-(ev - orecs['evals']))[:4]:
        print(f'    cand {i}: N={recs["n_pixels"][i]} M={recs["n_deform"][i]} evals GPU {recs["evals_full"][i]}+{recs["evals_value"][i]} (it {recs["iters_ell"][i]}/{recs["iters_dsm"][i]}) oracle {orecs["evals"][i]} (it {orecs["iters_ell"][i]}/{orecs["iters_dsm"][i]})  E GPU {recs["energy"][i]:.6g} oracle {orecs["energy"][i]:.6g} status {recs["status"][i]}/{orecs["status"][i]} flags {recs["flags"][i]}')
    for i in bad[:5]:
        print(f'    BAD cand {i}: N={recs["n_pixels"][i]} M={recs["n_deform"][i]} E GPU {recs["energy"][i]:.9g} oracle {orecs["energy"][i]:.9g} evals {ev[i]}/{orecs["evals"][i]}')
print('total evals GPU', tot_g, 'oracle', tot_o)
