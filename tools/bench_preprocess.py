#!/usr/bin/env python3
"""Times sdsm_preprocess (preprocess.py:39-68 on the GPU): ms per image and algorithmic GB/s (16 B/pixel: read g, write y)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superdsm_amd import engine, synth
for shape, sigma2 in (((520, 696), 10.0), ((1024, 1024), 42.43), ((4096, 4096), 10.0)):
    rng = np.random.default_rng(0)
    g = torch.as_tensor(rng.random(shape)).cuda()
    for _ in range(3):
        engine.preprocess(g, sigma2=sigma2, return_tensor=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        engine.preprocess(g, sigma2=sigma2, return_tensor=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    px = shape[0] * shape[1]
    print(f'{shape} sigma2={sigma2}: {dt * 1e3:.3f} ms per image, algorithmic {16 * px / dt / 1e9:.1f} GB/s ({16 * px / dt / 8e12 * 100:.2f} % of 8 TB/s)')
