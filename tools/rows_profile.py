"""Diagnostic (needs the -DSDSM_PROFILE build): where sdsm_k_setup_rows spends its time for the large regions of a workload -- the LAST
member's prologue, its runs, and meeting + envelope, in microseconds.   usage: SDSM_HIP_LIB=superdsm_amd/libsdsm_hip_prof.so python tools/rows_profile.py <workload> [images]"""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superdsm_amd import _capi, engine, testing
wl = sys.argv[1]
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 1
scenes = [testing.make_scene(wl, max_size=3, layout_index=k % 8 if wl == 'bbbc039_like' else 0) for k in range(nimg)]
fps = [fp for sc in scenes for fp in sc['footprints']]
image_of = np.concatenate([np.full(len(sc['footprints']), k, np.int32) for k, sc in enumerate(scenes)])
imgs = [engine.DeviceImage(sc['y'], None, sc['atoms'], sc['dsm_cfg']['background_margin']) for sc in scenes]
batch = engine.Batch(imgs if nimg > 1 else imgs[0], fps, scenes[0]['dsm_cfg'], image_of=image_of if nimg > 1 else None)
prof = torch.zeros(len(fps) * 24, dtype=torch.int64, device='cuda')
_capi.lib().sdsm_set_debug_buffer(C.c_void_p(prof.data_ptr()))
for _ in range(3):
    batch.launch()
torch.cuda.synchronize()
recs = batch.records()
ps = prof.cpu().numpy()[len(fps) * 16:].reshape(-1, 8).astype(np.float64) / 2400.0
big = np.argsort(-recs['n_pixels'])[:12]
for i in big:
    print('N=%6d M=%4d  rows kernel, last member, thread 0: loads of its runs %.0f  pass 1 (row sums) %.0f  pass 2 (entries) %.0f | setup kernel: sort + row table %.0f  row lengths + counting sort %.0f us | rows kernel, last member: prologue %.0f  runs %.0f  meeting + envelope %.0f us' % (recs['n_pixels'][i], recs['n_deform'][i], ps[i, 0], ps[i, 1], ps[i, 2], ps[i, 3], ps[i, 4], ps[i, 7], ps[i, 5], ps[i, 6]))
