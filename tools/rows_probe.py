import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superdsm_amd import _capi, engine, testing
scenes = [testing.make_scene('bbbc039_like', max_size=3, layout_index=k) for k in range(8)]
fps = [fp for sc in scenes for fp in sc['footprints']]
image_of = np.concatenate([np.full(len(sc['footprints']), k, np.int32) for k, sc in enumerate(scenes)])
imgs = [engine.DeviceImage(sc['y'], None, sc['atoms'], sc['dsm_cfg']['background_margin']) for sc in scenes]
batch = engine.Batch(imgs, fps, scenes[0]['dsm_cfg'], image_of=image_of)
prof = torch.zeros(len(fps) * 24, dtype=torch.int64, device='cuda')
_capi.lib().sdsm_set_debug_buffer(C.c_void_p(prof.data_ptr()))
for _ in range(3): batch.launch()
torch.cuda.synchronize()
a = prof.cpu().numpy()[16 * len(fps):16 * len(fps) + 6].copy()
batch.launch(); torch.cuda.synchronize()
b = prof.cpu().numpy()[16 * len(fps):16 * len(fps) + 6].copy()
d = (b - a).astype(float)
print('runs by thread 0 of the first 64 workgroups (setup + rows kernels):', int(d[4]), ' mean window of grid points scanned:', d[5] / max(d[4], 1))
for name, v in zip(('loads', 'pass 1 (row sums)', 'pass 2 (entries)', 'padding + meta'), d[:4]):
    print('  %-20s %.2f us per run' % (name, v / max(d[4], 1) / 100.0))
