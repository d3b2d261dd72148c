"""ctypes binding of the CPU oracle (oracle/sdsm_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing
under superdsm_amd/ does.  The oracle is the checker, never the thing measured as the product or shipped.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'liboracle.so')


def build(force=False):
    src = os.path.join(_HERE, 'sdsm_oracle.c')
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return _LIB_PATH


class DsmCfg(C.Structure):
    _fields_ = [('scale', C.c_double), ('epsilon', C.c_double), ('alpha', C.c_double), ('smooth_amount', C.c_double),
                ('gaussian_shape_multiplier', C.c_double), ('smooth_subsample', C.c_int), ('init_elliptical', C.c_int),
                ('max_iters', C.c_int), ('pad', C.c_int)]

    @staticmethod
    def from_dict(d):
        return DsmCfg(float(d.get('scale', 1000)), float(d.get('epsilon', 1.0)), float(d.get('alpha', 0.5)),
                      float(d.get('smooth_amount', 10)), float(d.get('gaussian_shape_multiplier', 2)),
                      int(d.get('smooth_subsample', 20)), int(d.get('init', 'elliptical') == 'elliptical'), int(d.get('max_iters', 100)), 0)


class CvxprogInfo(C.Structure):
    _fields_ = [('status', C.c_int), ('N', C.c_int), ('M', C.c_int), ('iters_ell', C.c_int), ('iters_dsm', C.c_int),
                ('evals', C.c_int), ('ell_status', C.c_int), ('retried', C.c_int), ('energy', C.c_double), ('energy_ell', C.c_double)]


class SolveInfo(C.Structure):
    _fields_ = [('status', C.c_int), ('iters', C.c_int), ('evals', C.c_int), ('value', C.c_double)]


RECORD_DTYPE = np.dtype([('energy', 'f8'), ('theta', 'f8', 6), ('status', 'i4'), ('is_optimal', 'i4'), ('on_boundary', 'i4'),
                         ('N', 'i4'), ('M', 'i4'), ('iters_ell', 'i4'), ('iters_dsm', 'i4'), ('evals', 'i4'),
                         ('fg_offset', 'i4', 2), ('fg_shape', 'i4', 2), ('frag_off', 'i8'), ('seconds', 'f8')], align=True)

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, f64 = C.c_void_p, C.c_int, C.c_double
        L.orc_gaussian_filter.argtypes = [vp, i32, i32, f64, vp]
        L.orc_gaussian_kernel1d.argtypes = [f64, i32, vp]
        L.orc_gaussian_radius.argtypes = [f64]
        L.orc_edt_sq.argtypes = [vp, i32, i32, vp]
        L.orc_preprocess.argtypes = [vp, i32, i32, f64, f64, f64, i32, vp]
        L.orc_bg_distance_sq.argtypes = [vp, i32, i32, vp]
        L.orc_region_mask.argtypes = [vp, vp, vp, i32, i32, vp, i32, f64, vp]
        L.orc_psf.argtypes = [f64, f64, vp]
        L.orc_smat_create.argtypes = [vp, i32, i32, f64, f64, i32]
        L.orc_smat_create.restype = vp
        L.orc_smat_free.argtypes = [vp]
        for fn in ('orc_smat_N', 'orc_smat_M'):
            getattr(L, fn).argtypes = [vp]
        L.orc_smat_nnz.argtypes = [vp]
        L.orc_smat_nnz.restype = C.c_int64
        L.orc_smat_shape.argtypes = [vp, vp]
        L.orc_smat_export.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_energy_create.argtypes = [vp, vp, i32, i32, f64, f64, i32, f64, f64, i32]
        L.orc_energy_create.restype = vp
        L.orc_energy_free.argtypes = [vp]
        L.orc_energy_N.argtypes = [vp]
        L.orc_energy_M.argtypes = [vp]
        L.orc_energy_smat.argtypes = [vp]
        L.orc_energy_smat.restype = vp
        L.orc_energy_eval.argtypes = [vp, vp, vp, vp]
        L.orc_energy_eval.restype = f64
        L.orc_newton.argtypes = [vp, f64, vp, vp, C.POINTER(SolveInfo)]
        L.orc_moment_init.argtypes = [vp, vp, i32, i32, vp]
        L.orc_cvxprog.argtypes = [vp, vp, i32, i32, C.POINTER(DsmCfg), vp, C.POINTER(CvxprogInfo)]
        L.orc_cvxprog.restype = vp
        L.orc_compute_objects.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, f64, C.POINTER(DsmCfg), i32]
        L.orc_compute_objects.restype = vp
        L.orc_set_inner_threads.argtypes = [i32]
        L.orc_set_exact_hessian.argtypes = [i32]
        L.orc_batch_free.argtypes = [vp]
        L.orc_batch_records.argtypes = [vp]
        L.orc_batch_records.restype = vp
        L.orc_batch_fragment.argtypes = [vp, i32]
        L.orc_batch_fragment.restype = vp
        L.orc_batch_params.argtypes = [vp, i32]
        L.orc_batch_params.restype = vp
        assert L.orc_record_size() == RECORD_DTYPE.itemsize, (L.orc_record_size(), RECORD_DTYPE.itemsize)
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def gaussian_filter(img, sigma):
    img = _f64(img)
    out = np.empty_like(img)
    lib().orc_gaussian_filter(_p(img), img.shape[0], img.shape[1], float(sigma), _p(out))
    return out


def edt_sq(nonzero):
    nz = _u8(nonzero)
    out = np.empty(nz.shape, np.int64)
    lib().orc_edt_sq(_p(nz), nz.shape[0], nz.shape[1], _p(out))
    return out


def preprocess(g, sigma1=np.sqrt(2), sigma2=40, offset_clip=3, lower_clip_mean=False):
    g = _f64(g)
    y = np.empty_like(g)
    lib().orc_preprocess(_p(g), g.shape[0], g.shape[1], float(sigma1), float(sigma2), float(offset_clip), int(lower_clip_mean), _p(y))
    return y


def region_mask(y, y_mask, atoms, labels, margin):
    y = _f64(y)
    atoms = np.ascontiguousarray(atoms, np.int32)
    d2 = np.empty(y.shape, np.int64)
    lib().orc_bg_distance_sq(_p(y), y.shape[0], y.shape[1], _p(d2))
    ym = _u8(y_mask) if y_mask is not None else None
    labels = np.ascontiguousarray(sorted(labels), np.int32)
    out = np.empty(y.shape, np.uint8)
    lib().orc_region_mask(_p(atoms), _p(ym) if ym is not None else None, _p(d2), y.shape[0], y.shape[1], _p(labels), len(labels), float(margin), _p(out))
    return out.astype(bool)


def psf(sigma, mult):
    k = lib().orc_psf(float(sigma), float(mult), None)
    out = np.empty((k, k), np.float32)
    lib().orc_psf(float(sigma), float(mult), _p(out))
    return out


class SmoothMatrix:
    def __init__(self, handle, owns=True):
        self.h, self.owns = handle, owns
        L = lib()
        self.N, self.M, nnz = L.orc_smat_N(handle), L.orc_smat_M(handle), L.orc_smat_nnz(handle)
        shp = np.zeros(3, np.int32)
        L.orc_smat_shape(handle, _p(shp))
        self.compressed_shape, self.k = (int(shp[0]), int(shp[1])), int(shp[2])
        self.indptr = np.zeros(self.N + 1, np.int64)
        self.indices = np.zeros(max(nnz, 1), np.int32)
        self.data = np.zeros(max(nnz, 1), np.float64)
        self.grid_r = np.zeros(max(self.M, 1), np.int32)
        self.grid_c = np.zeros(max(self.M, 1), np.int32)
        L.orc_smat_export(handle, _p(self.indptr), _p(self.indices), _p(self.data), _p(self.grid_r), _p(self.grid_c))
        self.indices, self.data = self.indices[:nnz], self.data[:nnz]
        self.grid_r, self.grid_c = self.grid_r[:self.M], self.grid_c[:self.M]

    def __del__(self):
        if self.owns and self.h:
            lib().orc_smat_free(self.h)
            self.h = None


def smooth_matrix(mask, sigma, mult, subsample):
    m = _u8(mask)
    return SmoothMatrix(lib().orc_smat_create(_p(m), m.shape[0], m.shape[1], float(sigma), float(mult), int(subsample)))


class Energy:
    """Restatement of superdsm.dsm.Energy on a full-image frame (region not shrunk)."""

    def __init__(self, y, mask, epsilon, alpha, smooth_amount=np.inf, mult=2, subsample=1):
        y, m = _f64(y), _u8(mask)
        deform = int(np.isfinite(smooth_amount))
        self.h = lib().orc_energy_create(_p(y), _p(m), y.shape[0], y.shape[1], float(epsilon), float(alpha), deform,
                                         float(smooth_amount) if deform else 0.0, float(mult), int(subsample) if deform else 1)
        self.N, self.M = lib().orc_energy_N(self.h), lib().orc_energy_M(self.h)
        self.n = 6 + self.M

    def __del__(self):
        if getattr(self, 'h', None):
            lib().orc_energy_free(self.h)
            self.h = None

    @property
    def smat(self):
        return SmoothMatrix(lib().orc_energy_smat(self.h), owns=False)

    def __call__(self, p):
        p = _f64(p)
        assert p.size == self.n
        return lib().orc_energy_eval(self.h, _p(p), None, None)

    def eval(self, p):
        p = _f64(p)
        assert p.size == self.n
        g = np.zeros(self.n)
        H = np.zeros((self.n, self.n))
        v = lib().orc_energy_eval(self.h, _p(p), _p(g), _p(H))
        return v, g, H

    def newton(self, x0, scale):
        x0 = _f64(x0)
        x = np.zeros(self.n)
        info = SolveInfo()
        lib().orc_newton(self.h, float(scale), _p(x0), _p(x), C.byref(info))
        return x, dict(status=info.status, iters=info.iters, evals=info.evals, value=info.value)


def moment_init(y, mask):
    y, m = _f64(y), _u8(mask)
    th = np.zeros(6)
    lib().orc_moment_init(_p(y), _p(m), y.shape[0], y.shape[1], _p(th))
    return th


def cvxprog(y, mask, dsm_cfg):
    y, m = _f64(y), _u8(mask)
    cfg = DsmCfg.from_dict(dsm_cfg)
    params = np.zeros(int(m.sum()) + 6)
    info = CvxprogInfo()
    h = lib().orc_cvxprog(_p(y), _p(m), y.shape[0], y.shape[1], C.byref(cfg), _p(params), C.byref(info))
    lib().orc_energy_free(h)
    out = {f: getattr(info, f) for f, _ in CvxprogInfo._fields_}
    return params[:6 + info.M].copy(), out


def compute_objects(y, y_mask, atoms, footprints, dsm_cfg, nthreads=0, inner_threads=1):
    """Returns (records structured array, list of bool fragments, list of parameter vectors).  nthreads: worker threads (one
    candidate each; 0 = all cores); inner_threads: threads per candidate (the passes over its pixels are split)."""
    y = _f64(y)
    atoms = np.ascontiguousarray(atoms, np.int32)
    ym = _u8(y_mask) if y_mask is not None else None
    offs = np.zeros(len(footprints) + 1, np.int32)
    offs[1:] = np.cumsum([len(fp) for fp in footprints])
    labels = np.ascontiguousarray(np.concatenate([sorted(fp) for fp in footprints]) if len(footprints) else np.zeros(0), np.int32)
    cfg = DsmCfg.from_dict(dsm_cfg)
    L = lib()
    L.orc_set_inner_threads(int(inner_threads))
    b = L.orc_compute_objects(_p(y), _p(ym) if ym is not None else None, _p(atoms), y.shape[0], y.shape[1], len(footprints),
                              _p(offs), _p(labels), float(dsm_cfg.get('background_margin', 20)), C.byref(cfg), int(nthreads))
    n = len(footprints)
    recs = np.ctypeslib.as_array(C.cast(L.orc_batch_records(b), C.POINTER(C.c_uint8)), shape=(n * RECORD_DTYPE.itemsize,)).view(RECORD_DTYPE).copy() if n else np.zeros(0, RECORD_DTYPE)
    frags, params = [], []
    for i in range(n):
        h, w = recs['fg_shape'][i]
        fp = L.orc_batch_fragment(b, i)
        frags.append(np.ctypeslib.as_array(C.cast(fp, C.POINTER(C.c_uint8)), shape=(h * w,)).reshape(h, w).astype(bool).copy())
        pp = L.orc_batch_params(b, i)
        params.append(np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_double)), shape=(6 + recs['M'][i],)).copy() if pp else None)
    L.orc_batch_free(b)
    L.orc_set_inner_threads(1)
    return recs, frags, params


def set_exact_hessian(on):
    """Cross-check mode: Newton on the reference's exact Hessian (no row threshold, no majoriser blend)."""
    lib().orc_set_exact_hessian(int(bool(on)))


def max_threads():
    return lib().orc_max_threads()
