#!/usr/bin/env python3
"""Start / end of every kernel of the last steps from the rocprofv3 kernel trace under gpurun_out/<tag>_trace -> profiles/<tag>_kernel_timeline.txt.
usage: python tools/kernel_timeline.py r02"""
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = glob.glob(os.path.join(root, 'gpurun_out', f'{tag}_trace', '*', '*kernel_trace.csv'))
rows = [r for r in csv.DictReader(open(max(files, key=os.path.getmtime))) if 'sdsm' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
last = [i for i, r in enumerate(rows) if 'sdsm_k_setup' in r['Kernel_Name'] and 'rows' not in r['Kernel_Name']][-4:]
lines = ['kernel trace of `python3 bench.py --no-cpu --no-extras --steps 6 --warmup 2 --repeats 2` (rocprofv3 --kernel-trace), times in ms since the first sdsm kernel;',
         'one step = sdsm_k_setup, then the size classes of sdsm_k_solve concurrently (class 1 does the work here; the lists of classes 2 and 3 are upper bounds whose workgroups exit at once and wait for free compute units)', '']
for r in rows[last[0]:]:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    lines.append(f'{name:62s} start {s:9.3f}  end {e:9.3f}  dur {e - s:7.3f}  grid {r.get("Grid_Size", r.get("Grid_Size_X", "?"))} wg {r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))}  '
                 f'vgpr {r.get("VGPR_Count", "?")} accum {r.get("Accum_VGPR_Count", "?")} sgpr {r.get("SGPR_Count", "?")} lds {r.get("LDS_Block_Size", "?")} scratch {r.get("Scratch_Size", "?")}')
open(os.path.join(root, 'profiles', f'{tag}_kernel_timeline.txt'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:12]))
