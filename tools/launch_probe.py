"""Diagnostic: launch_figures of bench.py for the NIH3T3-like image, alone and after other parts of the bench in the same process (the figure
in the full bench line is ~25 % above the one of `bench.py --workload nih3t3_like`)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import bench
from superdsm_amd import engine, testing
sys.argv = ['bench.py']
args = bench.parse()
nih = testing.make_scene('nih3t3_like', max_size=3)
show = lambda tag, f: print(tag, round(f['ms_per_launch'], 2), round(f['solve_kernels_ms'], 2), flush=True)
show('alone:', bench.launch_figures(nih))
scene = testing.make_scene('bbbc039_like', max_size=3)
img = engine.DeviceImage(scene['y'], None, scene['atoms'], scene['dsm_cfg']['background_margin'])
scenes = [testing.make_scene('bbbc039_like', max_size=3, layout_index=k) for k in range(8)]
ex = bench.extras(args, scene, img, 8, scenes)
show('after extras():', bench.launch_figures(nih))
oc = bench.other_configs(args)
print('inside other_configs:', round(oc['nih3t3_like']['ms_per_launch'], 2), round(oc['nih3t3_like']['solve_kernels_ms'], 2))
show('after other_configs():', bench.launch_figures(nih))
