#!/usr/bin/env python3
"""Diagnostic: duration of sdsm_batch_eval (one value pass + one full pass per candidate at the solution of a previous launch)
over the 8-image bench batch -- the cost of the passes in the throughput regime, without the solver around them.
usage: [SDSM_HIP_LIB=...] python tools/time_eval_pass.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superdsm_amd import _capi, engine, testing

scene = testing.make_scene('bbbc039_like', max_size=3)
n_images = 8
imgs = [engine.DeviceImage(scene['y'], None, scene['atoms'], scene['dsm_cfg']['background_margin']) for _ in range(n_images)]
fps = scene['footprints'] * n_images
image_of = np.repeat(np.arange(n_images, dtype=np.int32), len(scene['footprints']))
b = engine.Batch(imgs, fps, scene['dsm_cfg'], image_of=image_of, want_xi=True)
b.launch()
torch.cuda.synchronize()
L = _capi.lib()
npar, nout = L.sdsm_plan_eval_param_count(b.plan), L.sdsm_plan_eval_out_count(b.plan)
recs = b.records()
xi = b.xi_dev.cpu().numpy()
xo = b.xi_offsets()
buf = np.zeros(npar)
for i in range(len(fps)):
    m = int(recs['n_deform'][i])
    buf[6 * i + xo[i]:6 * i + xo[i] + 6] = recs['theta'][i]
    if m:
        buf[6 * i + xo[i] + 6:6 * i + xo[i] + 6 + m] = xi[xo[i]:xo[i] + m]
d_par = torch.from_numpy(buf).cuda()
d_out = torch.empty(nout, dtype=torch.float64, device='cuda')
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
ts = []
for r in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _capi.check(L.sdsm_batch_eval(b.plan, p(b.ws), b.ws_bytes, p(d_par), p(d_out), s), 'eval')
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print('sdsm_batch_eval over %d candidates: %s ms (median %.3f)' % (len(fps), ' '.join('%.3f' % t for t in ts), float(np.median(ts[2:]))))
npx = recs['n_pixels'].astype(np.int64)
print('pixels %d, sparse pixels %d' % (npx.sum(), npx[recs['n_deform'] > 0].sum()))
