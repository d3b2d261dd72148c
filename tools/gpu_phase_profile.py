#!/usr/bin/env python3
"""Diagnostic: per-candidate cycle counters of the solve kernel's phases (needs the -DSDSM_PROFILE build:
SDSM_HIP_LIB=superdsm_amd/libsdsm_hip_prof.so python tools/gpu_phase_profile.py [workload])."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superdsm_amd import _capi, engine, testing

wl = sys.argv[1] if len(sys.argv) > 1 else 'bbbc039_like'
scene = testing.make_scene(wl, max_size=3)
fps = scene['footprints']
img = engine.DeviceImage(scene['y'], None, scene['atoms'], scene['dsm_cfg']['background_margin'])
batch = engine.Batch(img, fps, scene['dsm_cfg'])
prof = torch.zeros(len(fps) * 24, dtype=torch.int64, device='cuda')     # 16 solve + 8 setup counters per candidate
_capi.lib().sdsm_set_debug_buffer(C.c_void_p(prof.data_ptr()))
for _ in range(2):
    batch.launch()
torch.cuda.synchronize()
recs = batch.records()
pall = prof.cpu().numpy()
p = pall[:len(fps) * 16].reshape(-1, 16).astype(np.float64)
ps = pall[len(fps) * 16:].reshape(-1, 8).astype(np.float64)
names = ['phaseA', 'phaseB', 'reduce', 'factor', 'linesrch', 'total', 'ell_tot']
n = recs['n_deform'] + 6
cls = np.where(n <= 40, 'A', np.where(n <= 84, 'B', np.where(n <= 172, 'C', 'D')))  # size bands for reporting only
print('cycles are shader-clock ticks of thread 0; ms = ticks / 2.4e6 (approx)')
for c in 'ABCD':
    m = cls == c
    if not m.any():
        continue
    print(f'class {c}: {m.sum()} candidates, N median {np.median(recs["n_pixels"][m]):.0f} max {recs["n_pixels"][m].max()}, M median {np.median(recs["n_deform"][m]):.0f} max {recs["n_deform"][m].max()}, '
          f'iters_dsm median {np.median(recs["iters_dsm"][m]):.0f} max {recs["iters_dsm"][m].max()}, full evals median {np.median(recs["evals_full"][m]):.0f}, value evals median {np.median(recs["evals_value"][m]):.0f}')
    tot = p[m, 5]
    print('   total ms: median %.3f  max %.3f  sum %.1f' % (np.median(tot) / 2.4e6, tot.max() / 2.4e6, tot.sum() / 2.4e6))
    for i, nm in enumerate(names[:5]):
        print('   %-9s share of total: %.1f%%   per full eval (median): %.1f us' % (nm, 100 * p[m, i].sum() / tot.sum(), np.median(p[m, i] / np.maximum(recs['evals_full'][m], 1)) / 2400))
    print('   elliptical share: %.1f%%' % (100 * p[m, 6].sum() / tot.sum()))
worst = np.argsort(-p[:, 5])[:8]
if p[:, 13:16].sum() > 0:
    m = cls == 'B'
    print('factor_solve parts, class B, us per call (median): head %.1f  panels %.1f  l2 + back substitution %.1f' % (np.median(p[m, 13] / np.maximum(recs['iters_dsm'][m], 1)) / 2400, np.median((p[m, 8:12].sum(1) + p[m, 14]) / np.maximum(recs['iters_dsm'][m], 1)) / 2400, np.median(p[m, 15] / np.maximum(recs['iters_dsm'][m], 1)) / 2400))
if p[:, 8:12].sum() > 0 and p[:, 13:16].sum() > 0:
    m = cls == 'B'
    tot = p[m, 8:12].sum()
    print('panel loop of the factorisation, class B, thread 0: diagonal block %.1f%%  trailing update %.1f%%  barrier wait %.1f%%  write-back %.1f%%' % tuple(100 * p[m, 8 + i].sum() / tot for i in range(4)))
if ps.sum() > 0:
    names_s = ['region scan', 'compressed coords + lattice', 'greedy grid', 'grid sort + row table', 'row lengths + counting sort', 'rows of G~ (2 passes)', 'envelope + state']
    print('setup kernel, thread 0, share of its time: ' + ',  '.join('%s %.1f%%' % (nm, 100 * ps[:, k].sum() / ps.sum()) for k, nm in enumerate(names_s)) + '   (median total %.0f us)' % (np.median(ps.sum(1)) / 2400))
if ps.sum() > 0:
    kbig = int(np.argmax(ps.sum(1)))
    print('slowest setup: cand %d N=%d M=%d total %.2f ms: ' % (kbig, recs['n_pixels'][kbig], recs['n_deform'][kbig], ps[kbig].sum() / 2.4e6) + ', '.join('%s %.2f' % (nm, ps[kbig, k] / 2.4e6) for k, nm in enumerate(names_s)))
print('slowest candidates:')
for k in worst:
    print('  cand %d N=%d M=%d it_ell=%d it_dsm=%d evals=%d/%d total=%.2f ms  A=%.2f B=%.2f red=%.2f fac=%.2f ls=%.2f' % (
        k, recs['n_pixels'][k], recs['n_deform'][k], recs['iters_ell'][k], recs['iters_dsm'][k], recs['evals_full'][k], recs['evals_value'][k],
        p[k, 5] / 2.4e6, *(p[k, :5] / 2.4e6)))
