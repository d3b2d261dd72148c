import ctypes as C, os, sys, numpy as np
sys.path.insert(0,'/root/repo')
import torch
from superdsm_amd import _capi, engine, testing
wl=sys.argv[1]
scene=testing.make_scene(wl,max_size=3)
fps=scene['footprints']
img=engine.DeviceImage(scene['y'],None,scene['atoms'],scene['dsm_cfg']['background_margin'])
batch=engine.Batch(img,fps,scene['dsm_cfg'])
prof=torch.zeros(len(fps)*24,dtype=torch.int64,device='cuda')
_capi.lib().sdsm_set_debug_buffer(C.c_void_p(prof.data_ptr()))
for _ in range(2): batch.launch()
torch.cuda.synchronize()
recs=batch.records()
ins=batch.inspect()
env=np.array([d['env_size'] for d in ins]); n=recs['n_deform']+6; N=recs['n_pixels']
p=prof.cpu().numpy()[:len(fps)*16].reshape(-1,16).astype(float)
tot=p[:,5]/2.4e6
k1=(n<=128)&(env<=2560); k2=~k1&(env<=11000); k3=~k1&~k2
for nm,m in (('K1',k1),('K2',k2),('K3',k3)):
    if m.any(): print(nm,'cands',m.sum(),'sum ms %.1f'%tot[m].sum(),'median ms %.2f'%np.median(tot[m]),'max ms %.2f'%tot[m].max(),'M median',int(np.median(recs['n_deform'][m])),'env median',int(np.median(env[m])),'env max',env[m].max(),'N median',int(np.median(N[m])))
print('env percentiles', np.percentile(env,[50,75,90,95,99,100]).astype(int), 'n percentiles', np.percentile(n,[50,75,90,95,99,100]).astype(int))
