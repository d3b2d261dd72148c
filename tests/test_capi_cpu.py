"""The C-ABI library loads on a CPU-only machine and exports every symbol include/sdsm.h declares; host-only entry
points work; compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'sdsm.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sdsm_[a-z_0-9]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound():
    from superdsm_amd import _capi
    lib = _capi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f'{name} is declared in include/sdsm.h but not exported by libsdsm_hip.so'
    assert sorted(_capi.SYMBOLS) == declared, 'ctypes binding table and header disagree'
    assert lib.sdsm_version() == 200


def test_record_and_config_layout():
    from superdsm_amd import _capi
    assert _capi.RECORD_DTYPE.itemsize == 128
    assert C.sizeof(_capi.DsmConfig) == 6 * 8 + 4 * 4
    cfg = _capi.make_config(dict(alpha=0.033, smooth_amount=4, smooth_subsample=8, background_margin=8))
    assert (cfg.scale, cfg.epsilon, cfg.init_elliptical, cfg.max_iters) == (1000.0, 1.0, 1, 100)


def test_host_only_entry_points():
    from superdsm_amd import _capi
    lib = _capi.lib()
    k = lib.sdsm_psf(4.0, 2.0, None)
    assert k == 33
    psf = np.zeros((k, k), np.float32)
    lib.sdsm_psf(4.0, 2.0, psf.ctypes.data_as(C.c_void_p))
    golden = np.load(os.path.join(ROOT, 'tests', 'golden', 'smoothmat_blob.npz'))['psf']
    np.testing.assert_array_equal(psf, golden)                 # bit exact float32 PSF of the reference (dsm.py:137-142,226)
    # planning is host-only: N and the mask boxes follow from the per-atom statistics
    stats = np.zeros((4, 6), np.int32)
    stats[1] = (10, 2, 4, 3, 6, 0)
    stats[2] = (7, 4, 5, 1, 3, 0)
    cfg = _capi.make_config(dict(alpha=0.033, smooth_amount=4, smooth_subsample=8, background_margin=8))
    offs = np.array([0, 1, 3], np.int32)
    labels = np.array([1, 1, 2], np.int32)
    plan = lib.sdsm_plan_create(40, 50, 3, stats.ctypes.data_as(C.c_void_p), C.byref(cfg), 2, offs.ctypes.data_as(C.c_void_p), labels.ctypes.data_as(C.c_void_p))
    assert plan
    info = np.zeros((2, 4), np.int32)
    moff = np.zeros(2, np.int64)
    npx = np.zeros(2, np.int32)
    assert lib.sdsm_plan_describe(plan, info.ctypes.data_as(C.c_void_p), moff.ctypes.data_as(C.c_void_p), npx.ctypes.data_as(C.c_void_p)) == 0
    assert npx.tolist() == [10, 17] and info.tolist() == [[2, 3, 3, 4], [2, 1, 4, 6]]
    assert lib.sdsm_plan_workspace_bytes(plan) > 0 and lib.sdsm_plan_mask_bytes(plan) >= 8
    lib.sdsm_plan_destroy(plan)
    bad = _capi.make_config(dict(alpha=-1.0))
    assert not lib.sdsm_plan_create(40, 50, 3, stats.ctypes.data_as(C.c_void_p), C.byref(bad), 0, None, None)
    assert b'alpha' in lib.sdsm_last_error()


def _plan_of_squares(sides):
    """A host-only plan over one candidate per atom, atoms = squares of the given side lengths on a grid (statistics only: area, extents)."""
    from superdsm_amd import _capi
    lib = _capi.lib()
    n = len(sides)
    pitch = max(sides)
    per_row = 60000 // pitch
    stats = np.zeros((n + 1, 6), np.int32)
    for k, sd in enumerate(sides, start=1):
        r, c = ((k - 1) // per_row) * pitch, ((k - 1) % per_row) * pitch
        stats[k] = (sd * sd, r, r + sd - 1, c, c + sd - 1, 0)
    H, W = ((n - 1) // per_row + 1) * pitch, per_row * pitch
    assert H <= 65535
    cfg = _capi.make_config(dict(alpha=0.033, smooth_amount=4, smooth_subsample=8, background_margin=0))
    offs = np.arange(n + 1, dtype=np.int32)
    labels = np.arange(1, n + 1, dtype=np.int32)
    plan = lib.sdsm_plan_create(H, W, n, stats.ctypes.data_as(C.c_void_p), C.byref(cfg), n, offs.ctypes.data_as(C.c_void_p), labels.ctypes.data_as(C.c_void_p))
    assert plan, lib.sdsm_last_error()
    return plan


def _schedule(plan, n, mode):
    from superdsm_amd import _capi
    lib = _capi.lib()
    assert lib.sdsm_plan_set_latency_mode(plan, mode) == 0
    g, r = np.zeros(n, np.int32), np.zeros(n, np.int32)
    assert lib.sdsm_plan_schedule(plan, g.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p)) == 0
    return g, r


def test_scheduling_policy_of_a_plan_is_host_logic():
    """Workgroup groups and the workgroups that build the rows of G~ (sdsm_api.hip, layout_plan), without a GPU: groups in throughput mode
    only for regions that are long next to the whole plan, the largest first within a budget of 256 members; latency mode groups the
    mid-size regions too; mode 2 none; rows by several workgroups only while regions above 4096 pixels are few."""
    from superdsm_amd import _capi
    lib = _capi.lib()
    # a few large regions among small ones (an image set with clusters): sides 30 (900 px) ... 180 (32 400 px)
    sides = [30] * 200 + [70] * 20 + [100] * 6 + [180] * 2
    n = len(sides)
    N = np.array(sides) ** 2
    plan = _plan_of_squares(sides)
    npx = np.zeros(n, np.int32)
    assert lib.sdsm_plan_describe(plan, None, None, npx.ctypes.data_as(C.c_void_p)) == 0
    assert (npx == N).all()
    g, r = _schedule(plan, n, 0)
    assert ((g > 0) == (N > 8192)).all() and (g[N > 8192] >= 2).all() and g.max() <= 8         # 10 000 px: 2 members, 32 400 px: 4
    assert ((r > 0) == (N > 4096)).all() and r[N == 4900].tolist() == [4] * 20                 # one workgroup per 1536 pixels
    g1, _ = _schedule(plan, n, 1)
    assert ((g1 > 0) == (N > 3072)).all() and (g1[N == 4900] == 3).all()                      # latency mode: one member per 2048 pixels, at most 4
    g2, r2 = _schedule(plan, n, 2)
    assert (g2 == 0).all() and ((r2 > 0) == (N > 4096)).all()
    lib.sdsm_plan_destroy(plan)
    # thousands of mid-size regions (the synthetic 4096^2 plan): the chip is full whatever the largest does -- no groups, rows inside the setup kernel
    sides = [70] * 8000 + [120] * 40
    n = len(sides)
    plan = _plan_of_squares(sides)
    g, r = _schedule(plan, n, 0)
    assert (g == 0).all() and (r == 0).all()
    lib.sdsm_plan_destroy(plan)
    # many large regions: the member budget goes to the largest (regions of equal size are treated alike: all or none)
    sides = list(range(120, 220)) + [100] * 100
    n = len(sides)
    N = np.array(sides) ** 2
    plan = _plan_of_squares(sides)
    for mode in (0, 1):
        g, _ = _schedule(plan, n, mode)
        assert 128 < g.sum() <= 256 and (g[N == 10000] == 0).all()
        smallest_grouped = N[g > 0].min()
        assert (g[N > smallest_grouped] > 0).all()
    lib.sdsm_plan_destroy(plan)


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from superdsm_amd import _capi, engine
    with pytest.raises(_capi.SdsmError):
        engine.DeviceImage(np.zeros((8, 8)), None, np.ones((8, 8), np.int32), 2)
    with pytest.raises(_capi.SdsmError):
        engine.preprocess(np.zeros((8, 8)))
