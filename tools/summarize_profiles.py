#!/usr/bin/env python3
"""Turns the rocprofv3 outputs merged under gpurun_out/ into the small committed summaries under profiles/.
usage: python tools/summarize_profiles.py r02 [workload] [images per launch]
The summary records the hash of the kernel sources it was measured on: bench.py reports its traffic figure only while that
hash matches the sources that are running."""
import collections
import csv
import glob
import json
import os
import hashlib
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
workload = sys.argv[2] if len(sys.argv) > 2 else 'bbbc039_like'
images = int(sys.argv[3]) if len(sys.argv) > 3 else 8
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, 'profiles'), exist_ok=True)
stats = glob.glob(os.path.join(root, 'gpurun_out', f'{tag}_trace', '*', '*kernel_stats.csv'))
if stats:
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(root, 'profiles', f'{tag}_kernel_stats.csv'))


def per_kernel(pattern, counter):
    files = glob.glob(os.path.join(root, 'gpurun_out', pattern, '*', '*counter_collection.csv'))
    agg = collections.defaultdict(lambda: [0, 0.0])
    if not files:
        return {}
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].split('(')[0][:80]
        agg[k][0] += 1
        agg[k][1] += float(r['Counter_Value'])
    return {k: dict(dispatches=n, avg_per_dispatch_KB=v / n) for k, (n, v) in agg.items() if 'sdsm' in k or 'k_' in k}


fetch = per_kernel(f'{tag}_fetch', 'FETCH_SIZE')
write = per_kernel(f'{tag}_write', 'WRITE_SIZE')
solve_fetch = sum(v['avg_per_dispatch_KB'] for k, v in fetch.items() if 'sdsm_k_solve' in k) * 1024
solve_write = sum(v['avg_per_dispatch_KB'] for k, v in write.items() if 'sdsm_k_solve' in k) * 1024
def source_hash():
    h = hashlib.sha1()
    d = os.path.join(root, 'superdsm_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


out = dict(
    source_hash=source_hash(), workload=workload, images_per_launch=images,
    note=('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --no-cpu --no-extras --steps 6 --warmup 2 --repeats 2`.  Counters are in KB.  '
          'MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a WIDE (16 B/lane) coalesced stream.  The bulk of what the solve '
          'kernels read is such a stream (the 16-byte weights of the run entries, global_load_dwordx4 at consecutive lanes; beside it the 16-byte halves of the '
          'packed crop and 4-byte index words): the reported traffic is 2 x FETCH_SIZE + WRITE_SIZE (since round 4; rounds 1-3 reported the raw counter, '
          'a lower bound); the raw value is kept beside it.'),
    fetch=fetch, write=write,
    solve_fetch_bytes_per_launch_raw=solve_fetch, solve_fetch_bytes_per_launch_x2=2 * solve_fetch,
    solve_write_bytes_per_launch=solve_write,
    solve_hbm_bytes_per_launch=2 * solve_fetch + solve_write, solve_hbm_bytes_per_launch_raw=solve_fetch + solve_write)
sq = {}
files = glob.glob(os.path.join(root, 'gpurun_out', f'{tag}_sq', '*', '*counter_collection.csv'))
if files:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        if 'sdsm_k_solve<128' in r['Kernel_Name']:
            agg['solve class 1'][r['Counter_Name']].append(float(r['Counter_Value']))
        elif 'sdsm_k_setup' in r['Kernel_Name']:
            agg['setup'][r['Counter_Name']].append(float(r['Counter_Value']))
    sq = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
    for k, d in sq.items():
        wc = d.get('SQ_WAVE_CYCLES', 0)
        if wc:
            d['frac_wave_parked_on_waitcnt_or_barrier'] = d.get('SQ_WAIT_ANY', 0) / wc
            d['frac_issue_stalled'] = d.get('SQ_WAIT_INST_ANY', 0) / wc
            d['frac_issuing'] = d.get('SQ_ACTIVE_INST_ANY', 0) / wc
        if d.get('SQ_LDS_IDX_ACTIVE'):
            d['lds_bank_conflict_share_of_lds_cycles'] = d.get('SQ_LDS_BANK_CONFLICT', 0) / d['SQ_LDS_IDX_ACTIVE']
out['sq_counters_per_launch'] = sq
out['sq_note'] = 'separate pass: --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; averages per launch'
json.dump(out, open(os.path.join(root, 'profiles', f'{tag}_pmc_summary.json'), 'w'), indent=1)
print(json.dumps({k: out[k] for k in out if k.startswith('solve_')}, indent=1))
