"""Diagnostic (needs the -DSDSM_PROFILE build): candidates, workgroup time and sizes per solve class of one launch.
usage: SDSM_HIP_LIB=superdsm_amd/libsdsm_hip_prof.so python tools/class_stats.py <workload>"""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
wide=np.array([d.get('wide_g',0) for d in ins]) if 'wide_g' in ins[0] else np.zeros(len(fps),int)
k1=(n<=128)&(env<=2560); k1b=~k1&(n<=256)&(env<=6144); k2=~k1&~k1b&(env<=11000); k2b=~k1&~k1b&~k2&(n<=512)&(env<=15900); k3=~k1&~k1b&~k2&~k2b
grp=(N>12288)&(env<=11000)          # throughput mode: regions of more than 12 288 pixels whose envelope fits class 2 are solved by workgroup groups
for nm,m in (('K1',k1&~grp),('K1b',k1b&~grp),('K2',k2&~grp),('K2b',k2b),('K3',k3),('groups',grp)):
    if m.any(): print(nm,'cands',m.sum(),'sum ms %.1f'%tot[m].sum(),'median ms %.2f'%np.median(tot[m]),'max ms %.2f'%tot[m].max(),'M median',int(np.median(recs['n_deform'][m])),'M max',int(recs['n_deform'][m].max()),'env median',int(np.median(env[m])),'env max',env[m].max(),'N median',int(np.median(N[m])),'N max',int(N[m].max()))
print('env percentiles', np.percentile(env,[50,75,90,95,99,100]).astype(int), 'n percentiles', np.percentile(n,[50,75,90,95,99,100]).astype(int))
