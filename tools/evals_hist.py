import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from superdsm_amd import testing
tot = []
for k in range(8):
    sc = testing.make_scene('bbbc039_like', max_size=3, layout_index=k)
    recs = testing.solve_scene_gpu(sc)['records']
    tot.append(recs)
r = np.concatenate(tot)
ef, ev, N, M = r['evals_full'], r['evals_value'], r['n_pixels'], r['n_deform']
w = (ef + 0.5 * ev) * N          # rough cost: passes x pixels
print('candidates', len(r), 'evals_full percentiles 10/50/90/99/max', np.percentile(ef, [10, 50, 90, 99]), ef.max())
for lo, hi in ((0, 20), (20, 30), (30, 40), (40, 60), (60, 100), (100, 100000)):
    m = (ef >= lo) & (ef < hi)
    print('evals_full in [%d, %d): %5d candidates, %5.1f %% of pass-pixels, median N %d, median M %d, iters_dsm median %d' % (lo, hi, m.sum(), 100 * w[m].sum() / w.sum(), np.median(N[m]) if m.any() else 0, np.median(M[m]) if m.any() else 0, np.median(r['iters_dsm'][m]) if m.any() else 0))
print('iters_ell percentiles', np.percentile(r['iters_ell'], [10, 50, 90, 99]), 'iters_dsm', np.percentile(r['iters_dsm'], [10, 50, 90, 99]))
print('status counts', np.unique(r['status'], return_counts=True), 'flags&1 (second elliptical attempt):', int((r['flags'] & 1).sum()))
