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


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from superdsm_amd import _capi, engine
    with pytest.raises(_capi.SdsmError):
        engine.DeviceImage(np.zeros((8, 8)), None, np.ones((8, 8), np.int32), 2)
    with pytest.raises(_capi.SdsmError):
        engine.preprocess(np.zeros((8, 8)))
