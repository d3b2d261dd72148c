"""Diagnostic: a digest of the 128-byte records of every candidate of the bench workloads -- two builds that claim the same results
bit for bit (SDSM_HIP_LIB=... python tools/record_digest.py) must print the same lines."""
import hashlib, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superdsm_amd import testing
for wl, layouts in (('bbbc039_like', range(8)), ('gowt1_like', [0]), ('nih3t3_like', [0]), ('synthetic4096', [0])):
    if len(sys.argv) > 1 and wl not in sys.argv[1:]:
        continue
    for k in layouts:
        sc = testing.make_scene(wl, layout_index=k)
        recs = testing.solve_scene_gpu(sc)['records']
        print(wl, k, len(recs), hashlib.sha1(np.ascontiguousarray(recs).tobytes()).hexdigest(), 'evals', int(recs['evals_full'].sum()), int(recs['evals_value'].sum()), flush=True)
