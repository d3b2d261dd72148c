"""``postprocess`` stage: discards spurious objects and refines the masks (reference: superdsm/postprocess.py:13-344).

Same stage name, inputs (``cover, y_img, atoms, g_raw, dsm_cfg``), output (``postprocessed_objects``), hyper-parameters and
``configure_ex`` factors as the reference.  The reference hands every object of the cover to a Ray task that runs two
Euclidean distance transforms of the WHOLE image (contrast, postprocess.py:254-266; the region of the normalised energy,
:289-291).  Here the per-object work of an image is ONE batch of the HIP engine (sdsm_post_objects: contrast response and
mask refinement on a window around each object; the two smoothed images by the separable Gaussian kernels); the normalised
energy needs no recomputation (``Object.cvxprog_region_size``).  What stays on the host, on the small fragments: hole
filling, the glare test, the eccentricity and the accept / discard decisions.  There is no CPU path for the batch."""
import math
import os

import numpy as np
import scipy.ndimage as ndi

from . import _capi, _morph
from .objects import BaseObject
from .output import get_output
from .pipeline import Stage


class PostprocessedObject(BaseObject):
    """A segmented object after post-processing (postprocess.py:244-251)."""

    def __init__(self, original):
        self.original = original
        self.fg_offset = original.fg_offset
        self.fg_fragment = original.fg_fragment


def _is_glare(obj, g_smooth, min_layer=0.5, num_layers=5):
    """Top ``1 - min_layer`` of the smoothed intensity profile connected at every of ``num_layers`` levels (postprocess.py:269-286)."""
    h, w = obj.fg_fragment.shape
    sect = g_smooth[obj.fg_offset[0]:obj.fg_offset[0] + h, obj.fg_offset[1]:obj.fg_offset[1] + w]
    mask = _morph.binary_erosion(obj.fg_fragment, _morph.disk(2))
    data = sect[mask]
    for prop in np.linspace(min_layer, 1, num_layers, endpoint=False):
        layer = np.logical_and(mask, sect > (data.max() - data.min()) * prop + data.min())
        if ndi.label(layer)[0].max() > 1:
            return False
    return True


def _compute_eccentricity(fragment):
    """Eccentricity of the ellipse with the region's second moments (what ``skimage.measure.regionprops(...).eccentricity`` is,
    postprocess.py:340-344): sqrt(1 - l2 / l1) of the eigenvalues l1 >= l2 of the inertia tensor."""
    if not fragment.any():
        return 0
    rr, cc = np.nonzero(fragment)
    r, c = rr - rr.mean(), cc - cc.mean()
    a, b, d = (r * r).mean(), (r * c).mean(), (c * c).mean()
    half, root = (a + d) / 2, math.sqrt(((a - d) / 2) ** 2 + b * b)
    l1, l2 = half + root, half - root
    return 0.0 if l1 == 0 else math.sqrt(max(0.0, 1 - l2 / l1))


def gaussian_filter_gpu(g_dev, sigma):
    """scipy.ndimage.gaussian_filter(g, sigma) on the device (a float64 H x W tensor in, a new tensor out)."""
    import ctypes as C
    import torch
    L = _capi.lib()
    H, W = (int(v) for v in g_dev.shape)
    out = torch.empty_like(g_dev)
    nbytes = L.sdsm_gaussian_workspace_bytes(H, W, float(sigma))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=g_dev.device)
    with torch.cuda.device(g_dev.device):
        _capi.check(L.sdsm_gaussian_filter(C.c_void_p(g_dev.data_ptr()), H, W, float(sigma), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes,
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'sdsm_gaussian_filter')
    return out


def process_objects_gpu(objects, g, g_mask_processing, background_mask, exterior_scale, exterior_offset, contrast_epsilon,
                        mask_max_distance, mask_stdamp, device=None):
    """Contrast response and refined mask (before hole filling) of every object: one launch (sdsm_post_objects).
    ``g`` / ``g_mask_processing``: float64 device tensors; ``background_mask``: bool array or uint8 device tensor.
    Returns (records POST_RECORD_DTYPE, list of (offset, fragment) or None where the device did not refine)."""
    import ctypes as C
    import torch
    from . import engine
    L = _capi.lib()
    dev = g.device
    H, W = (int(v) for v in g.shape)
    n = len(objects)
    if n == 0:
        return np.zeros(0, _capi.POST_RECORD_DTYPE), []
    boxes = np.zeros((n, 4), np.int32)
    words = np.zeros(n, np.int64)
    if mask_max_distance > 0 and mask_stdamp > 0 and float(mask_max_distance) != int(mask_max_distance):
        # skimage.morphology.disk(r) of a fractional radius (postprocess.py:316-337) is not a disk of int(r): refuse instead of truncating
        raise NotImplementedError(f'mask_max_distance = {mask_max_distance!r}: the GPU mask refinement takes integer radii (<= 16) only, see DESIGN.md "Limits"')
    m = int(mask_max_distance) if (mask_max_distance > 0 and mask_stdamp > 0) else 0
    new_words = np.zeros(n, np.int64)
    packed = []
    for k, obj in enumerate(objects):
        h, w = obj.fg_fragment.shape
        r0, c0 = int(obj.fg_offset[0]), int(obj.fg_offset[1])
        boxes[k] = (r0, c0, h, w)
        bits = np.packbits(np.ascontiguousarray(obj.fg_fragment, bool).reshape(-1), bitorder='little')
        nw = (h * w + 31) // 32
        buf = np.zeros(nw * 4, np.uint8)
        buf[:bits.size] = bits
        packed.append(buf)
        words[k] = nw
        nh = min(H, r0 + h + m) - max(0, r0 - m)
        nwid = min(W, c0 + w + m) - max(0, c0 - m)
        new_words[k] = (nh * nwid + 31) // 32
    bits_off = np.concatenate([[0], np.cumsum(words)[:-1]]).astype(np.int64)
    new_off = np.concatenate([[0], np.cumsum(new_words)[:-1]]).astype(np.int64)
    # objects with a very long mask boundary keep their boundary list in global memory (4 B per boundary pixel <= mask pixels)
    areas = np.array([int(o.fg_fragment.sum()) for o in objects], np.int64)
    need_pool = areas > 12288
    bpool_off = np.where(need_pool, np.concatenate([[0], np.cumsum(np.where(need_pool, areas, 0))[:-1]]), -1).astype(np.int64)
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_boxes, d_bits_off, d_new_off, d_bpool_off = to_dev(boxes), to_dev(bits_off), to_dev(new_off), to_dev(bpool_off)
    d_bits = to_dev(np.concatenate(packed))
    d_new = torch.zeros(max(1, int(new_words.sum())) * 4, dtype=torch.uint8, device=dev)
    d_pool = torch.empty(max(1, int(np.where(need_pool, areas, 0).sum())) * 4, dtype=torch.uint8, device=dev)
    d_out = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    bg = background_mask if torch.is_tensor(background_mask) else to_dev(np.asarray(background_mask, np.uint8))
    gstd = float(g.std(unbiased=False).item())               # a constant image: 1 / 0 = inf, the contrast comes out NaN and nothing is discarded, as in the reference (postprocess.py:254-266)
    p = lambda t: C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        _capi.check(L.sdsm_post_objects(p(g), p(g_mask_processing), p(bg), H, W, n, p(d_boxes), p(d_bits_off), p(d_bits), p(d_new_off), p(d_new),
                                        p(d_pool), p(d_bpool_off), float(exterior_scale), float(exterior_offset), float(contrast_epsilon),
                                        (1.0 / gstd) if gstd > 0 else float('inf'), m, float(mask_stdamp), p(d_out), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                    'sdsm_post_objects')
        recs = d_out.cpu().numpy().view(_capi.POST_RECORD_DTYPE).copy()
        new_bits = d_new.cpu().numpy() if m > 0 else None
    if (recs['status'] == 1).any():
        raise _capi.SdsmError('sdsm_post_objects: boundary list overflow')
    refined = []
    for k in range(n):
        if m == 0:
            refined.append(None)
            continue
        r0, c0, h, w = (int(v) for v in boxes[k])
        nr0, nc0 = max(0, r0 - m), max(0, c0 - m)
        nh, nwid = min(H, r0 + h + m) - nr0, min(W, c0 + w + m) - nc0
        if recs['h'][k] <= 0:
            refined.append((np.zeros(2, int), np.zeros((1, 1), bool)))            # extract_foreground_fragment of an empty mask
            continue
        win = np.unpackbits(new_bits[4 * new_off[k]:4 * (new_off[k] + new_words[k])], bitorder='little')[:nh * nwid].reshape(nh, nwid).astype(bool)
        fr, fc = int(recs['r0'][k]) - nr0, int(recs['c0'][k]) - nc0
        refined.append((np.array([int(recs['r0'][k]), int(recs['c0'][k])]), win[fr:fr + int(recs['h'][k]), fc:fc + int(recs['w'][k])]))
    return recs, refined


class Postprocessing(Stage):
    """Stage ``postprocess`` (hyper-parameters as documented in superdsm/postprocess.py:13-110)."""

    ENABLED_BY_DEFAULT = True

    def __init__(self):
        super().__init__('postprocess', inputs=['cover', 'y_img', 'atoms', 'g_raw', 'dsm_cfg'], outputs=['postprocessed_objects'])

    def process(self, input_data, cfg, out, log_root_dir):
        import torch
        out = get_output(out)
        # simple post-processing
        max_norm_energy = cfg.get('max_norm_energy', 0.2)
        discard_image_boundary = cfg.get('discard_image_boundary', False)
        min_boundary_obj_radius = cfg.get('min_boundary_obj_radius', 0)
        min_obj_radius = cfg.get('min_object_radius', 0)
        max_obj_radius = cfg.get('max_object_radius', np.inf)
        max_eccentricity = cfg.get('max_eccentricity', 0.99)
        max_boundary_eccentricity = cfg.get('max_boundary_eccentricity', np.inf)
        if max_boundary_eccentricity is None:
            max_boundary_eccentricity = max_eccentricity
        # contrast-based post-processing
        exterior_scale = cfg.get('exterior_scale', 5)
        exterior_offset = cfg.get('exterior_offset', 5)
        min_contrast = cfg.get('min_contrast', 1.35)
        contrast_epsilon = cfg.get('contrast_epsilon', 1e-4)
        # mask-based post-processing
        mask_stdamp = cfg.get('mask_stdamp', 2)
        mask_max_distance = cfg.get('mask_max_distance', 1)
        mask_smoothness = cfg.get('mask_smoothness', 3)
        fill_holes = cfg.get('fill_holes', True)
        # autofluorescence glare removal
        glare_detection_smoothness = cfg.get('glare_detection_smoothness', 3)
        glare_detection_num_layers = cfg.get('glare_detection_num_layers', 5)
        glare_detection_min_layer = cfg.get('glare_detection_min_layer', 0.5)
        min_glare_radius = cfg.get('min_glare_radius', np.inf)
        min_boundary_glare_radius = cfg.get('min_boundary_glare_radius', min_glare_radius)

        g_raw = np.asarray(input_data['g_raw'], np.float64)
        solution = list(input_data['cover'].solution)
        # pixels allowed for the background estimate of the contrast (postprocess.py:152-155)
        background_mask = np.zeros(g_raw.shape, bool)
        for c in solution:
            c.fill_foreground(background_mask)
        background_mask = _morph.binary_erosion(~background_mask, _morph.disk(exterior_offset))

        # (the reference's filter reads the loop variable of the loop above, postprocess.py:180: all objects or none)
        objects = [obj for obj in solution if (solution[-1].fg_fragment.any() if solution else False)]
        g_dev = torch.as_tensor(np.ascontiguousarray(g_raw)).cuda()
        g_mask = gaussian_filter_gpu(g_dev, mask_smoothness)
        recs, refined = process_objects_gpu(objects, g_dev, g_mask, background_mask, exterior_scale, exterior_offset, contrast_epsilon,
                                            mask_max_distance, mask_stdamp)
        need_glare = any((min_boundary_glare_radius if o.on_boundary else min_glare_radius) < math.sqrt(o.fg_fragment.sum() / math.pi) for o in objects)
        g_glare = gaussian_filter_gpu(g_dev, glare_detection_smoothness).cpu().numpy() if need_glare else None

        postprocessed_objects, log_entries = [], []
        for k, original in enumerate(objects):
            obj_radius = math.sqrt(original.fg_fragment.sum() / math.pi)
            is_glare = False
            if (min_boundary_glare_radius if original.on_boundary else min_glare_radius) < obj_radius:
                is_glare = _is_glare(original, g_glare, glare_detection_min_layer, glare_detection_num_layers)
            norm_energy = original.energy / original.cvxprog_region_size                    # postprocess.py:289-291 without the second distance transform
            contrast_response = float(recs['contrast'][k])
            if refined[k] is not None:                                                       # postprocess.py:316-337
                fg_offset, fg_fragment = refined[k]
                if fill_holes:
                    fg_fragment = ndi.binary_fill_holes(fg_fragment)
            elif fill_holes:
                fg_offset, fg_fragment = original.fg_offset, ndi.binary_fill_holes(original.fg_fragment)
            else:
                fg_offset, fg_fragment = None, None
            eccentricity = _compute_eccentricity(original.fg_fragment)

            obj = PostprocessedObject(original)
            if fg_fragment is not None and fg_offset is not None:
                obj.fg_fragment, obj.fg_offset = fg_fragment.copy(), np.asarray(fg_offset).copy()
                if not obj.fg_fragment.any():
                    log_entries.append((obj, 'empty foreground'))
                    continue
            if is_glare:
                log_entries.append((obj, f'glare removed (radius: {obj_radius})'))
                continue
            if norm_energy > max_norm_energy:
                log_entries.append((obj, f'energy rate too high ({norm_energy})'))
                continue
            if contrast_response < min_contrast:
                log_entries.append((obj, f'contrast too low ({contrast_response})'))
                continue
            if original.on_boundary:
                if eccentricity > max_boundary_eccentricity:
                    log_entries.append((obj, f'boundary object eccentricity too high ({eccentricity})'))
                    continue
                if discard_image_boundary:
                    log_entries.append((obj, 'boundary object discarded'))
                    continue
                if not (min_boundary_obj_radius <= obj_radius <= max_obj_radius):
                    log_entries.append((obj, f'boundary object and/or too small/large (radius: {obj_radius})'))
                    continue
            else:
                if eccentricity > max_eccentricity:
                    log_entries.append((obj, f'eccentricity too high ({eccentricity})'))
                    continue
                if not min_obj_radius <= obj_radius <= max_obj_radius:
                    log_entries.append((obj, f'object too small/large (radius: {obj_radius})'))
                    continue
            postprocessed_objects.append(obj)

        if log_root_dir is not None:
            os.makedirs(log_root_dir, exist_ok=True)
            with open(os.path.join(log_root_dir, 'postprocessing.txt'), 'w') as log_file:
                for c, comment in log_entries:
                    location = (c.fg_offset + np.divide(c.fg_fragment.shape, 2)).round().astype(int)
                    log_file.write(f'object at x={location[1]}, y={location[0]}: {comment}{os.linesep}')
        out.write(f'Remaining objects: {len(postprocessed_objects)} of {len(objects)}')
        self.last_records = recs
        return {'postprocessed_objects': postprocessed_objects}

    def configure_ex(self, scale, radius, diameter):
        return {
            'min_object_radius': (radius, 0.0),
            'max_object_radius': (radius, np.inf),
            'min_glare_radius': (radius, np.inf),
        }
