"""Diagnostic: counters of the device-built class work lists after a launch of the 8-layout plan."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superdsm_amd import _capi, engine, testing
nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scenes = [testing.make_scene('bbbc039_like', max_size=3, layout_index=k % 8) for k in range(nimg)]
fps = [fp for sc in scenes for fp in sc['footprints']]
image_of = np.concatenate([np.full(len(sc['footprints']), k, np.int32) for k, sc in enumerate(scenes)])
imgs = [engine.DeviceImage(sc['y'], None, sc['atoms'], sc['dsm_cfg']['background_margin']) for sc in scenes]
batch = engine.Batch(imgs, fps, scenes[0]['dsm_cfg'], image_of=image_of)
batch.launch(); torch.cuda.synchronize()
ws = batch.ws.cpu().numpy()
n = batch.n
al = lambda v: (v + 255) // 256 * 256
off_list = batch.ws_bytes - al(4 * 4 * n)
blk = ws[off_list - 256:off_list].view(np.int32)
print('ticket', blk[0], 'counts', blk[16:20], 'heads', blk[24:28])
lists = ws[off_list:off_list + 4 * 4 * n].view(np.int32).reshape(4, n)
for l in range(4):
    print('list', l, lists[l, :max(0, min(12, blk[16 + l]))])
