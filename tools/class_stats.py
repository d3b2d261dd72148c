"""Diagnostic (needs the -DSDSM_PROFILE build): candidates, workgroup time and sizes per solve class of one launch.
usage: SDSM_HIP_LIB=superdsm_amd/libsdsm_hip_prof.so python tools/class_stats.py <workload> [images]
(bbbc039_like with images > 1: one plan over that many DIFFERENT BBBC039-like images, as bench.py's step)"""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superdsm_amd import _capi, engine, testing
wl = sys.argv[1]
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 1
same = len(sys.argv) > 3 and sys.argv[3] == 'same'      # the step of rounds 1-2: copies of ONE image
scenes = [testing.make_scene(wl, max_size=3, layout_index=k % 8 if wl == 'bbbc039_like' and not same else 0) for k in range(nimg)]
fps = [fp for sc in scenes for fp in sc['footprints']]
image_of = np.concatenate([np.full(len(sc['footprints']), k, np.int32) for k, sc in enumerate(scenes)])
imgs = [engine.DeviceImage(sc['y'], None, sc['atoms'], sc['dsm_cfg']['background_margin']) for sc in scenes]
batch = engine.Batch(imgs if nimg > 1 else imgs[0], fps, scenes[0]['dsm_cfg'], image_of=image_of if nimg > 1 else None)
prof = torch.zeros(len(fps) * 24, dtype=torch.int64, device='cuda')
_capi.lib().sdsm_set_debug_buffer(C.c_void_p(prof.data_ptr()))
for _ in range(2):
    batch.launch()
torch.cuda.synchronize()
_capi.lib().sdsm_enable_kernel_timing(1)
batch.launch()
print('solve kernels of the launch: %.2f ms, setup %.2f ms' % (_capi.lib().sdsm_last_solve_kernel_ms(), _capi.lib().sdsm_last_setup_kernel_ms()))
recs = batch.records()
lay = np.zeros(16, np.int64)
_capi.check(_capi.lib().sdsm_plan_layout(batch.plan, lay.ctypes.data_as(C.c_void_p)), 'layout')
ws = batch.ws.cpu().numpy()
state = ws[lay[1]:lay[1] + int(lay[14]) * batch.n].view(np.int32).reshape(batch.n, -1)
env = state[:, 15].astype(np.int64); n = recs['n_deform'] + 6; N = recs['n_pixels']
p = prof.cpu().numpy()[:len(fps) * 16].reshape(-1, 16).astype(float)
tot = p[:, 5] / 2.4e6
k1 = (n <= 128) & (env <= 2560); k1b = ~k1 & (n <= 256) & (env <= 7168); k2 = ~k1 & ~k1b & (env <= 11000); k2b = ~k1 & ~k1b & ~k2 & (n <= 512) & (env <= 15170); k3 = ~k1 & ~k1b & ~k2 & ~k2b
gm = np.zeros(batch.n, np.int32)
_capi.check(_capi.lib().sdsm_plan_schedule(batch.plan, gm.ctypes.data_as(C.c_void_p), None), 'schedule')
grp = (gm > 0) & (((n <= 1024) & (env <= 11000)) | ((n <= 512) & (env <= 15170)))     # solved by a workgroup group (the plan's own schedule, sdsm_plan_schedule, and the layouts of sdsm_solve_class)
for nm, m in (('K1', k1 & ~grp), ('K1b', k1b & ~grp), ('K2', k2 & ~grp), ('K2b', k2b), ('K3', k3), ('groups', grp)):
    if m.any():
        print(nm, 'cands', m.sum(), 'sum ms %.1f' % tot[m].sum(), 'median ms %.2f' % np.median(tot[m]), 'max ms %.2f' % tot[m].max(), 'M median', int(np.median(recs['n_deform'][m])), 'M max', int(recs['n_deform'][m].max()),
              'env median', int(np.median(env[m])), 'env max', env[m].max(), 'N median', int(np.median(N[m])), 'N max', int(N[m].max()))
print('env percentiles', np.percentile(env, [50, 75, 90, 95, 99, 100]).astype(int), 'n percentiles', np.percentile(n, [50, 75, 90, 95, 99, 100]).astype(int))
print('slowest candidates (ms, N, M, env, evals full/value, phases A / factor / line search):')
for k in np.argsort(-tot)[:12]:
    print('  %.2f  N=%d M=%d env=%d evals=%d/%d  A=%.2f fac=%.2f ls=%.2f' % (tot[k], N[k], recs['n_deform'][k], env[k], recs['evals_full'][k], recs['evals_value'][k], p[k, 0] / 2.4e6, p[k, 3] / 2.4e6, p[k, 4] / 2.4e6))
print('factor_solve parts of the slowest candidates, ms: head | diag block, trailing update, barrier wait, write-back (panel loop, thread 0) | after loop | norm + back substitution')
for k in np.argsort(-tot)[:6]:
    print('  N=%d M=%d: %.2f | %.2f %.2f %.2f %.2f | %.2f | %.2f   (factorisations: %d)' % (N[k], recs['n_deform'][k], p[k, 13] / 2.4e6, p[k, 8] / 2.4e6, p[k, 9] / 2.4e6, p[k, 10] / 2.4e6, p[k, 11] / 2.4e6,
                                                                                          p[k, 14] / 2.4e6, p[k, 15] / 2.4e6, recs['evals_full'][k]))

# when the workgroups of the launch started and ended (100 MHz wall clock, relative to the first start): who ends the launch
st, en = p[:, 7] / 1e5, p[:, 12] / 1e5
ok = (st > 0) & (en > 0)
t0 = st[ok].min()
st, en = st - t0, en - t0
print('launch span (first workgroup start to last end): %.2f ms' % en[ok].max())
cls = np.where(grp, 5, np.where(k1, 0, np.where(k1b, 1, np.where(k2, 2, np.where(k2b, 3, 4)))))
names = ['K1', 'K1b', 'K2', 'K2b', 'K3', 'groups']
for c in range(6):
    m = ok & (cls == c)
    if m.any():
        print('  %-6s first start %.2f  last start %.2f  last end %.2f   workgroup-ms started in [0,1) [1,2) [2,3) [3,4) [4,5) [5,..): %s' % (
            names[c], st[m].min(), st[m].max(), en[m].max(), ' '.join('%.0f' % tot[m & (st >= a) & (st < b)].sum() for a, b in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 99)))))
print('the last 15 to end (end, start, duration ms, class, N, M):')
for k in np.argsort(-np.where(ok, en, -1))[:15]:
    print('  end %.2f start %.2f dur %.2f %s N=%d M=%d' % (en[k], st[k], en[k] - st[k], names[cls[k]], N[k], recs['n_deform'][k]))
# residency: workgroups of class 1 running at time t
for t in (0.25, 0.5, 1, 2, 3, 4, 5, 6):
    print('  t=%.2f ms: running K1 %d  K1b %d  others %d' % (t, int((ok & (cls == 0) & (st <= t) & (en > t)).sum()), int((ok & (cls == 1) & (st <= t) & (en > t)).sum()), int((ok & (cls > 1) & (st <= t) & (en > t)).sum())))
