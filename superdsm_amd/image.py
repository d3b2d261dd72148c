"""Image / region carrier of the hot path (reference: superdsm/image.py:6-103), restated."""
import numpy as np


def get_pixel_map(shape, normalized=False):
    """Row and column coordinate arrays of an array of ``shape`` (stacked); divided by ``shape - 1`` if
    ``normalized`` (image.py:19-21: a singleton axis is left undivided)."""
    shape = tuple(int(s) for s in shape)
    grids = np.indices(shape).astype(float)
    if normalized:
        for axis, extent in enumerate(shape):
            if extent > 1:
                grids[axis] /= extent - 1.0
    return grids


def bbox(mask, include_end=False):
    """Bounding box ``[[r0, r1], [c0, c1]]`` of a mask and the matching slice (image.py:24-45)."""
    rows = np.flatnonzero(mask.any(axis=1))
    cols = np.flatnonzero(mask.any(axis=0))
    box = np.array([[rows[0], rows[-1]], [cols[0], cols[-1]]])
    if not include_end:
        box[:, 1] += 1
    return box, np.s_[box[0][0]:box[0][1], box[1][0]:box[1][1]]


def normalize_image(img):
    """Min-max normalisation to [0, 1] (image.py:48-57); a constant image maps to zeros."""
    lo, hi = img.min(), img.max()
    span = hi - lo
    return (img - lo).astype(float) / (span if span != 0 else 1)


class Image:
    """``model`` (intensities), ``mask`` (where the model is valid), ``offset`` of a shrunk region."""

    def __init__(self, model=None, mask=None, full_mask=None, offset=(0, 0)):
        self.model = model
        self.mask = np.ones(model.shape, bool) if mask is None else mask
        self.full_mask = self.mask if full_mask is None else full_mask
        self.offset = offset

    @staticmethod
    def create_from_array(img, mask=None, normalize=True):
        assert mask is None or (isinstance(mask, np.ndarray) and mask.dtype == bool)
        return Image(model=normalize_image(img) if normalize else img, mask=mask)

    def shrink_mask(self, mask):
        r, c = self.offset
        return mask[r:r + self.mask.shape[0], c:c + self.mask.shape[1]]

    def get_region(self, mask, shrink=False):
        mask = np.logical_and(self.mask, mask)
        if not shrink:
            return Image(self.model, mask)                     # region NOT shrunk: full-image frame (image.py:87)
        box, sel = bbox(mask)
        return Image(self.model[sel], mask[sel], full_mask=mask, offset=tuple(box[:, 0]))

    def get_map(self, normalized=True, pad=0):
        assert pad >= 0 and isinstance(pad, int)
        return get_pixel_map(np.add(self.model.shape, 2 * pad), normalized)
