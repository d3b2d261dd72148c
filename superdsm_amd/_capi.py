"""ctypes binding of the C ABI declared in include/sdsm.h (libsdsm_hip.so).

There is no CPU fallback: if the shared library is missing or a symbol cannot be resolved, importing callers
fail loudly here.  PyTorch-ROCm is used only as the owner of device memory (tensors -> raw device pointers)
and of the stream.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SDSM_HIP_LIB', os.path.join(_HERE, 'libsdsm_hip.so'))   # override: diagnostic builds only

SDSM_OK = 0
ATOM_STATS_STRIDE = 6

CAND_OPTIMAL, CAND_FALLBACK, CAND_TRIVIAL, CAND_ERROR, CAND_UNSUPPORTED, CAND_GIVEN_UP = 0, 1, 2, 3, 4, 5


class DsmConfig(C.Structure):
    """sdsm_dsm_config: DSM_CONFIG_DEFAULTS of the reference (superdsm/dsmcfg.py:6-21)."""
    _fields_ = [('scale', C.c_double), ('epsilon', C.c_double), ('alpha', C.c_double), ('smooth_amount', C.c_double),
                ('gaussian_shape_multiplier', C.c_double), ('background_margin', C.c_double),
                ('smooth_subsample', C.c_int32), ('init_elliptical', C.c_int32), ('max_iters', C.c_int32), ('flags', C.c_int32)]


RECORD_DTYPE = np.dtype([
    ('energy', 'f8'), ('theta', 'f8', 6), ('energy_ell', 'f8'),
    ('status', 'i4'), ('flags', 'i4'), ('n_pixels', 'i4'), ('n_deform', 'i4'),
    ('iters_ell', 'i4'), ('iters_dsm', 'i4'), ('evals_value', 'i4'), ('evals_full', 'i4'),
    ('on_boundary', 'i4'), ('fg_r0', 'i4'), ('fg_c0', 'i4'), ('fg_h', 'i4'), ('fg_w', 'i4'), ('n_positive', 'i4'), ('n_negative', 'i4'), ('reserved', 'i4')])
assert RECORD_DTYPE.itemsize == 128
POST_RECORD_DTYPE = np.dtype([('contrast', 'f8'), ('interior_mean', 'f8'), ('exterior_mean', 'f8'), ('fg_mean', 'f8'), ('fg_std', 'f8'),
                              ('area', 'i4'), ('status', 'i4'), ('r0', 'i4'), ('c0', 'i4'), ('h', 'i4'), ('w', 'i4')])
assert POST_RECORD_DTYPE.itemsize == 64

# every entry point of include/sdsm.h: name -> (restype, argtypes)
_vp, _i32, _f64, _sz, _i64 = C.c_void_p, C.c_int, C.c_double, C.c_size_t, C.c_int64
SYMBOLS = {
    'sdsm_version': (_i32, []),
    'sdsm_last_error': (C.c_char_p, []),
    'sdsm_device_count': (_i32, []),
    'sdsm_set_device': (_i32, [_i32]),
    'sdsm_stream_synchronize': (_i32, [_vp]),
    'sdsm_psf': (_i32, [_f64, _f64, _vp]),
    'sdsm_preprocess_workspace_bytes': (_sz, [_i32, _i32, _f64, _f64]),
    'sdsm_preprocess': (_i32, [_vp, _i32, _i32, _f64, _f64, _f64, _i32, _vp, _vp, _sz, _vp]),
    'sdsm_image_workspace_bytes': (_sz, [_i32, _i32]),
    'sdsm_image_prepare': (_i32, [_vp, _vp, _vp, _i32, _i32, _f64, _i32, _vp, _vp, _vp, _sz, _vp]),
    'sdsm_plan_create': (_vp, [_i32, _i32, _i32, _vp, C.POINTER(DsmConfig), _i32, _vp, _vp]),
    'sdsm_plan_create_multi': (_vp, [_i32, _vp, _vp, _vp, _vp, C.POINTER(DsmConfig), _i32, _vp, _vp, _vp]),
    'sdsm_plan_destroy': (None, [_vp]),
    'sdsm_plan_workspace_bytes': (_sz, [_vp]),
    'sdsm_plan_mask_bytes': (_sz, [_vp]),
    'sdsm_plan_describe': (_i32, [_vp, _vp, _vp, _vp]),
    'sdsm_plan_total_pixels': (_i64, [_vp]),
    'sdsm_batch_upload': (_i32, [_vp, _vp, _sz, _vp]),
    'sdsm_batch_launch': (_i32, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    'sdsm_batch_launch_multi': (_i32, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    'sdsm_plan_xi_count': (_i64, [_vp]),
    'sdsm_plan_xi_offsets': (_i32, [_vp, _vp]),
    'sdsm_plan_layout': (_i32, [_vp, _vp]),
    'sdsm_plan_schedule': (_i32, [_vp, _vp, _vp]),
    'sdsm_plan_set_latency_mode': (_i32, [_vp, _i32]),
    'sdsm_post_objects': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _f64, _i32, _f64, _vp, _vp]),
    'sdsm_gaussian_workspace_bytes': (_sz, [_i32, _i32, _f64]),
    'sdsm_gaussian_filter': (_i32, [_vp, _i32, _i32, _f64, _vp, _vp, _sz, _vp]),
    'sdsm_separable_workspace_bytes': (_sz, [_i32, _i32, _i32, _i32]),
    'sdsm_separable_filter': (_i32, [_vp, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _sz, _vp]),
    'sdsm_minsetcover': (_i32, [_i32, _i32, _vp, _vp, _f64, _i32, _i32, _f64, _vp, _vp]),
    'sdsm_minsetcover_multi': (_i32, [_i32, _vp, _vp, _vp, _vp, _f64, _i32, _i32, _f64, _vp, _vp]),
    'sdsm_maxsetpack': (_i32, [_i32, _i32, _vp, _vp, _vp, _vp]),
    'sdsm_count_growth': (_i32, [_i32, _vp, _vp, _vp, _i32, _i64, _vp]),
    'sdsm_unpack_fragments': (_i64, [_vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    'sdsm_plan_eval_param_count': (_i64, [_vp]),
    'sdsm_plan_eval_out_count': (_i64, [_vp]),
    'sdsm_batch_eval': (_i32, [_vp, _vp, _sz, _vp, _vp, _vp]),
    'sdsm_batch_deform_counts': (_i32, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'sdsm_plan_set_start': (_i32, [_vp, _vp]),
    'sdsm_enable_kernel_timing': (_i32, [_i32]),
    'sdsm_last_solve_kernel_ms': (_f64, []),
    'sdsm_last_setup_kernel_ms': (_f64, []),
    'sdsm_set_debug_buffer': (_i32, [_vp]),
    'sdsm_set_group_timeout_us': (_i32, [_f64]),
    'sdsm_side_queues_distinct': (_i32, []),
}

_lib = None


class SdsmError(RuntimeError):
    pass


def lib():
    """Loads libsdsm_hip.so and resolves every symbol of include/sdsm.h; raises if anything is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SdsmError(f'{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950). '
                            'There is no CPU fallback for the DSM solve path.')
        # PyTorch-ROCm owns the device memory and ships its own HIP runtime (same SONAME as /opt/rocm's): it must be
        # loaded first so that this library binds to the SAME runtime instance instead of bringing up a second one.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        assert L.sdsm_version() >= 200
        _lib = L
    return _lib


def check(code, what=''):
    if code != SDSM_OK:
        raise SdsmError(f'{what} failed ({code}): {lib().sdsm_last_error().decode()}')


def make_config(dsm_cfg):
    """dict with the reference's ``dsm/*`` keys (dsmcfg.py:6-21) -> sdsm_dsm_config."""
    d = dict(dsm_cfg)
    sa = d.get('smooth_amount', 10)
    return DsmConfig(scale=float(d.get('scale', 1000)), epsilon=float(d.get('epsilon', 1.0)), alpha=float(d.get('alpha', 0.5)),
                     smooth_amount=float(sa), gaussian_shape_multiplier=float(d.get('gaussian_shape_multiplier', 2)),
                     background_margin=float(d.get('background_margin', 20)), smooth_subsample=int(d.get('smooth_subsample', 20)),
                     init_elliptical=int(not callable(d.get('init', 'elliptical')) and d.get('init', 'elliptical') == 'elliptical'), max_iters=int(d.get('max_iters', 100)), flags=1 if d.get('no_trivial_rule') else 0)
