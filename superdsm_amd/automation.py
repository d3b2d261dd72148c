"""AF_ expansion of scale-dependent hyper-parameters and automatic scale estimation (reference: superdsm/automation.py).

``create_config`` is automation.py:80-102.  ``_estimate_scale`` restates automation.py:41-68: determinant-of-Hessian blobs on
11 scales (radii 20 … 200 pixels), restricted to pixels where the Laplacian of Gaussian is negative; the scale is the mean
radius of the inlier blobs / sqrt(2).  The Laplacian-of-Gaussian masks -- separable filters with up to 1100 taps over the whole
image, by far the largest part of the reference's cost -- run on the GPU (sdsm_separable_filter with SciPy's own
derivative-of-Gaussian weights); the box-filter determinant, the 3x3x3 peak search and the overlap pruning are small array
operations on the host.  PARITY UNPINNED: the reference takes ``_hessian_matrix_det``, ``peak_local_max`` and ``_prune_blobs``
from scikit-image, which is not available to this build; they are restated from their documented behaviour (box filters on
the integral image as in SURF; local maxima of a 3x3x3 neighbourhood above the threshold; of two blobs overlapping by more
than ``overlap`` the smaller one is dropped) and checked on synthetic images of known scale only."""
import math

import numpy as np
import scipy.ndimage as ndi
from scipy.spatial import cKDTree


def normalize_image(img, spread=1):
    """Contrast enhancement (superdsm/render.py:137-165)."""
    img = np.asarray(img, np.float64)
    if not np.allclose(img.std(), 0):
        minval, maxval = max(img.min(), img.mean() - spread * img.std()), min(img.max(), img.mean() + spread * img.std())
        img = img.clip(minval, maxval)
    img = img - img.min()
    return img / img.max()


def _integ(ii, r, c, rl, cl):
    """Clamped box sum on the integral image (rows r .. r + rl, columns c .. c + cl as scikit-image's ``_integ`` takes them):
    the origin is clipped FIRST and the far corner is measured from the clipped origin, so a window that starts above / left of
    the image slides into it instead of shrinking (parity unpinned, see the module docstring)."""
    H, W = ii.shape
    r1, c1 = np.clip(r, 0, H - 1), np.clip(c, 0, W - 1)
    r2, c2 = np.clip(r1 + rl, 0, H - 1), np.clip(c1 + cl, 0, W - 1)
    ans = ii[r1[:, None], c1[None, :]] + ii[r2[:, None], c2[None, :]] - ii[r1[:, None], c2[None, :]] - ii[r2[:, None], c1[None, :]]
    return np.maximum(0, ans)


def _hessian_matrix_det(ii, sigma):
    """Approximate determinant of the Hessian at scale ``sigma`` from box filters on the integral image ``ii``."""
    size = int(3 * sigma)
    H, W = ii.shape
    s2, s3, w = (size - 1) // 2, size // 3, size
    w_i = 1.0 / size / size
    r, c = np.arange(H), np.arange(W)
    tl = _integ(ii, r - s3, c - s3, s3, s3)
    br = _integ(ii, r + 1, c + 1, s3, s3)
    bl = _integ(ii, r - s3, c + 1, s3, s3)
    tr = _integ(ii, r + 1, c - s3, s3, s3)
    dxy = -(bl + tr - tl - br) * w_i
    mid = _integ(ii, r - s3 + 1, c - s2, 2 * s3 - 1, w)
    side = _integ(ii, r - s3 + 1, c - s3 // 2, 2 * s3 - 1, s3)
    dxx = -(mid - 3 * side) * w_i
    mid = _integ(ii, r - s2, c - s3 + 1, w, 2 * s3 - 1)
    side = _integ(ii, r - s3 // 2, c - s3 + 1, s3, 2 * s3 - 1)
    dyy = -(mid - 3 * side) * w_i
    return dxx * dyy - 0.81 * (dxy * dxy)


def _disk_overlap(d, r1, r2):
    ratio1 = np.clip((d ** 2 + r1 ** 2 - r2 ** 2) / (2 * d * r1), -1, 1)
    ratio2 = np.clip((d ** 2 + r2 ** 2 - r1 ** 2) / (2 * d * r2), -1, 1)
    a, b, c, e = -d + r2 + r1, d - r2 + r1, d + r2 - r1, d + r2 + r1
    area = r1 ** 2 * math.acos(ratio1) + r2 ** 2 * math.acos(ratio2) - 0.5 * math.sqrt(abs(a * b * c * e))
    return area / (math.pi * min(r1, r2) ** 2)


def _blob_overlap(b1, b2):
    root = math.sqrt(2)
    if b1[-1] == b2[-1] == 0:
        return 0.0
    if b1[-1] > b2[-1]:
        max_sigma, r1, r2 = b1[-1], 1.0, b2[-1] / b1[-1]
    else:
        max_sigma, r1, r2 = b2[-1], b1[-1] / b2[-1], 1.0
    p1, p2 = b1[:2] / (max_sigma * root), b2[:2] / (max_sigma * root)
    d = math.sqrt(float(((p2 - p1) ** 2).sum()))
    if d > r1 + r2:
        return 0.0
    if d <= abs(r1 - r2):
        return 1.0
    return _disk_overlap(d, r1, r2)


def _prune_blobs(blobs, overlap):
    """Of two blobs whose disks (radius sigma * sqrt(2)) overlap by more than ``overlap`` of the smaller, drop the smaller."""
    sigma = blobs[:, -1].max()
    pairs = sorted(cKDTree(blobs[:, :-1]).query_pairs(2 * sigma * math.sqrt(blobs.shape[1] - 1)))
    for i, j in pairs:
        if _blob_overlap(blobs[i], blobs[j]) > overlap:
            if blobs[i][-1] > blobs[j][-1]:
                blobs[j][-1] = 0
            else:
                blobs[i][-1] = 0
    return blobs[blobs[:, -1] > 0]


def _blob_doh(image, sigma_list, threshold=0.01, overlap=.5, mask=None):
    """Determinant-of-Hessian blobs (automation.py:13-38): rows (r, c, sigma)."""
    image = np.asarray(image, np.float64)
    if mask is None:
        mask = np.ones(image.shape, bool)
    if not isinstance(mask, dict):
        mask = {sigma: mask for sigma in sigma_list}
    ii = image.cumsum(0).cumsum(1)
    cube = np.dstack([mask[s] * _hessian_matrix_det(ii, s) for s in sigma_list])
    peaks = (cube == ndi.maximum_filter(cube, footprint=np.ones((3, 3, 3)), mode='nearest')) & (cube > threshold)
    if peaks.all():
        peaks[:] = False
    coords = np.transpose(np.nonzero(peaks))
    if coords.size == 0:
        return np.empty((0, 3))
    coords = coords[np.argsort(-cube[tuple(coords.T)], kind='stable')]
    lm = coords.astype(np.float64)
    lm[:, -1] = np.asarray(sigma_list)[coords[:, -1]]
    return _prune_blobs(lm, overlap)


def _log_negative_masks(im, sigma_list):
    """``{sigma: ndi.gaussian_laplace(im, sigma) < 0}`` (automation.py:52) with the filters on the GPU.  The weights are SciPy's own
    (``_gaussian_kernel1d`` of order 0 and 2: both symmetric), the axes are filtered in SciPy's order."""
    import ctypes as C
    import torch
    from scipy.ndimage._filters import _gaussian_kernel1d
    from . import _capi
    L = _capi.lib()
    H, W = im.shape
    d_in = torch.as_tensor(np.ascontiguousarray(im, np.float64)).cuda()
    a, b = torch.empty_like(d_in), torch.empty_like(d_in)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    out = {}
    for sigma in sigma_list:
        R = int(4.0 * float(sigma) + 0.5)
        w0 = np.ascontiguousarray(_gaussian_kernel1d(float(sigma), 0, R)[::-1])
        w2 = np.ascontiguousarray(_gaussian_kernel1d(float(sigma), 2, R)[::-1])
        nbytes = L.sdsm_separable_workspace_bytes(H, W, R, R)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=d_in.device)
        hp = lambda w: w.ctypes.data_as(C.c_void_p)
        _capi.check(L.sdsm_separable_filter(p(d_in), H, W, hp(w2), R, hp(w0), R, p(a), p(ws), nbytes, stream), 'sdsm_separable_filter')   # d2/dr2
        _capi.check(L.sdsm_separable_filter(p(d_in), H, W, hp(w0), R, hp(w2), R, p(b), p(ws), nbytes, stream), 'sdsm_separable_filter')   # d2/dc2
        out[sigma] = ((a + b) < 0).cpu().numpy()
    return out


def _estimate_scale(im, min_radius=20, max_radius=200, num_radii=10, thresholds=[0.01], inlier_tol=np.inf):
    """Estimates the scale sigma of the objects of an image (automation.py:41-68).  Returns (scale, blobs, inlier mask)."""
    sigma_list = np.linspace(min_radius, max_radius, num_radii) / math.sqrt(2)
    sigma_list = np.concatenate([[sigma_list.min() / 2], sigma_list])
    im_norm = normalize_image(im)
    im_norm /= im_norm.max()
    blobs_mask = _log_negative_masks(im_norm, sigma_list)
    mean_radius = None
    for threshold in sorted(thresholds, reverse=True):
        blobs_doh = _blob_doh(im_norm, sigma_list, threshold=threshold, mask=blobs_mask)
        blobs_doh = blobs_doh[~np.isclose(blobs_doh[:, 2], sigma_list.min())]
        if len(blobs_doh) == 0:
            continue
        radii = blobs_doh[:, 2] * math.sqrt(2)
        radii_median = np.median(radii)
        radii_mad = np.mean(np.abs(radii - np.median(radii)))
        radii_inliers = np.logical_and(radii >= radii_median - radii_mad, radii <= radii_median + radii_mad)
        mean_radius = np.mean(radii[radii_inliers])
        break
    if mean_radius is None:
        raise ValueError('scale estimation failed')
    return mean_radius / math.sqrt(2), blobs_doh, radii_inliers


def _expand(cfg, key, factor, default_user_factor, type=None, min=None, max=None):
    *ns, leaf = key.split('/')
    af_key = '/'.join(ns + ['AF_' + leaf])
    cfg.set_default(key, factor * cfg.get(af_key, default_user_factor), True)
    if type is not None:
        cfg.update(key, func=type)
    if min is not None:
        cfg.update(key, func=lambda value: value if value >= min else min)
    if max is not None:
        cfg.update(key, func=lambda value: value if value <= max else max)


def create_config(pipeline, base_cfg, img=None):
    """Scale-dependent hyper-parameters from ``AF_scale`` or, if that is not set, from the estimated scale of ``img``
    (automation.py:80-102)."""
    cfg = base_cfg.copy()
    scale = cfg.get('AF_scale', None)
    if scale is None:
        if img is None:
            raise ValueError('AF_scale is not set and there is no image to estimate the scale from')
        scale = _estimate_scale(img, num_radii=10, thresholds=[0.01])[0]
    for stage in pipeline.stages:
        for key, spec in stage.configure(scale).items():
            assert len(spec) in (2, 3), f'{type(stage).__name__}.configure returned tuple of unknown length ({len(spec)})'
            _expand(cfg, f'{stage.cfgns}/{key}', spec[0], spec[1], **(spec[2] if len(spec) == 3 else {}))
    return cfg, scale
