"""CPU restatement of the per-object work of the reference's post-processing stage.  TEST INFRASTRUCTURE ONLY (see
oracle/oracle.py): only tests/ may import it; nothing under superdsm_amd/ does.

Follows superdsm/postprocess.py in the reference's own FULL-IMAGE formulation (one Euclidean distance transform of the whole
image per object), so that it is an independent check of the windowed GPU kernel:
    compute_contrast      postprocess.py:254-266
    process_mask          postprocess.py:316-337
    is_glare              postprocess.py:269-286
    compute_eccentricity  postprocess.py:340-344 -> skimage.measure.regionprops(...).eccentricity
Pinned by tests/golden/postprocess.npz (outputs of the reference's own functions; its scikit-image calls -- disk,
binary_dilation, binary_erosion -- ran on their SciPy equivalents, tests/golden/_refshim.py).  NOT pinned: the eccentricity
(scikit-image is absent; restated from its documented definition: sqrt(1 - l2 / l1) of the eigenvalues l1 >= l2 of the inertia
tensor of the region's pixels)."""
import numpy as np
import scipy.ndimage as ndi


def disk(r):
    d = np.arange(-int(r), int(r) + 1)
    return (d[:, None] ** 2 + d[None, :] ** 2) <= int(r) ** 2


def binary_dilation(img, se):
    return ndi.binary_dilation(img, structure=se)


def binary_erosion(img, se):
    return ndi.binary_erosion(img, structure=se, border_value=True)


def paste(shape, offset, fragment):
    out = np.zeros(shape, bool)
    out[offset[0]:offset[0] + fragment.shape[0], offset[1]:offset[1] + fragment.shape[1]] = fragment
    return out


def extract_fragment(mask):
    rows = np.flatnonzero(mask.any(axis=1))
    if rows.size == 0:
        return np.zeros(2, int), np.zeros((1, 1), bool)
    cols = np.flatnonzero(mask.any(axis=0))
    return np.array([rows[0], cols[0]]), mask[rows[0]:rows[-1] + 1, cols[0]:cols[-1] + 1]


def background_mask(shape, objects, exterior_offset):
    """postprocess.py:152-155; objects: list of (offset, fragment)."""
    fg = np.zeros(shape, bool)
    for off, frag in objects:
        fg[off[0]:off[0] + frag.shape[0], off[1]:off[1] + frag.shape[1]] = frag       # fill_foreground ASSIGNS the whole box (objects.py:44-47): a later object's box overwrites
    return binary_erosion(~fg, disk(exterior_offset))


def compute_contrast(offset, fragment, g, exterior_scale, exterior_offset, epsilon, bg_mask):
    g = g / g.std()
    mask = paste(g.shape, offset, fragment)
    interior_mean = g[mask].mean()
    dmap = (ndi.distance_transform_edt(~mask) - exterior_offset).clip(0, np.inf) / exterior_scale
    ext = np.logical_xor(mask, dmap <= 5) & bg_mask
    w = np.zeros(g.shape)
    w[ext] = np.exp(-dmap[ext])
    w /= w.sum()
    exterior_mean = (g * w).sum()
    return (interior_mean + epsilon) / (exterior_mean + epsilon)


def process_mask(offset, fragment, g_smooth, max_distance, stdamp, fill_holes=False):
    if stdamp <= 0 or max_distance <= 0:
        return (offset, ndi.binary_fill_holes(fragment)) if fill_holes else (None, None)
    mask = paste(g_smooth.shape, offset, fragment)
    superset = np.logical_xor(binary_dilation(mask, disk(max_distance)), binary_erosion(mask, disk(max_distance)))
    data = g_smooth[mask]
    mean, amp = data.mean(), data.std() * stdamp
    extra_fg = (mean - amp <= g_smooth) & (g_smooth <= mean + amp)
    mask[superset & extra_fg] = True
    mask[superset & ~extra_fg] = False
    off, frag = extract_fragment(mask)
    if fill_holes:
        frag = ndi.binary_fill_holes(frag)
    return off, frag


def is_glare(offset, fragment, g_smooth, min_layer=0.5, num_layers=5):
    sect = g_smooth[offset[0]:offset[0] + fragment.shape[0], offset[1]:offset[1] + fragment.shape[1]]
    mask = binary_erosion(fragment, disk(2))
    data = sect[mask]
    for prop in np.linspace(min_layer, 1, num_layers, endpoint=False):
        layer = mask & (sect > (data.max() - data.min()) * prop + data.min())
        if ndi.label(layer)[0].max() > 1:
            return False
    return True


def compute_eccentricity(fragment):
    if not fragment.any():
        return 0
    rr, cc = np.nonzero(fragment)
    r, c = rr - rr.mean(), cc - cc.mean()
    a, b, d = (r * r).mean(), (r * c).mean(), (c * c).mean()
    half, root = (a + d) / 2, np.sqrt(((a - d) / 2) ** 2 + b * b)
    l1, l2 = half + root, half - root
    return 0.0 if l1 == 0 else float(np.sqrt(max(0.0, 1 - l2 / l1)))
