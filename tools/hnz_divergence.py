#!/usr/bin/env python3
"""Diagnostic: how much of the Hessian work of the sparse pixel pass is paid for by exec-masked lanes.  A wavefront of the solve
kernel walks the leading ('Hessian') entries of 64 consecutive crop positions in lock step: its cost follows the LARGEST count
among them.  Prints, per workload, the sum over wavefronts of work(max count) against the sum over pixels of work(count),
work(h) = h (h + 1) / 2 + 6 h atomics.  usage: python tools/hnz_divergence.py [workload]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superdsm_amd import engine, image, objects, testing

wl = sys.argv[1] if len(sys.argv) > 1 else 'bbbc039_like'
scene = testing.make_scene(wl, max_size=3)
y = image.Image.create_from_array(scene['y'], normalize=False)
cfg = dict(scene['dsm_cfg'])
margin = cfg.pop('background_margin', 20)
im = objects.device_image(y, scene['atoms'], margin)
b = engine.Batch(im, scene['footprints'], cfg)
b.launch()
recs, _ = b.download()
work = lambda h: h * (h + 1) // 2 + 6 * h
tot_lane = tot_wave = tot_sorted = 0
hist = np.zeros(32, np.int64)
gl = gw = 0
for c in b.inspect():
    h = c['hnz_stored_order']
    z = c['nnz_stored_order']
    if c['M'] == 0 or len(h) == 0:
        continue
    hist += np.bincount(np.minimum(h, 31), minlength=32)
    n = len(h)
    pad = (-n) % 64
    hw = np.concatenate([h, np.zeros(pad, h.dtype)]).reshape(-1, 64)
    tot_lane += work(h).sum()
    tot_wave += 64 * work(hw.max(axis=1)).sum()
    # ideal: positions of one group-count class sorted by their count
    g = (z + 3) // 4
    o = np.lexsort((-h, -g))
    hs = np.concatenate([h[o], np.zeros(pad, h.dtype)]).reshape(-1, 64)
    tot_sorted += 64 * work(hs.max(axis=1)).sum()
    zw = np.concatenate([z, np.zeros(pad, z.dtype)]).reshape(-1, 64)
    gl += z.sum(); gw += 64 * zw.max(axis=1).sum()
print(f'{wl}: leading-entry count histogram {hist[:16].tolist()}')
print(f'Hessian atomics: per lane {tot_lane / 1e6:.2f} M, as issued per wavefront (x64) {tot_wave / 1e6:.2f} M = {tot_wave / tot_lane:.2f}x; '
      f'with positions sorted by count inside a group-count class {tot_sorted / 1e6:.2f} M = {tot_sorted / tot_lane:.2f}x')
print(f'gradient atomics: per lane {gl / 1e6:.2f} M, per wavefront {gw / 1e6:.2f} M = {gw / gl:.2f}x')
