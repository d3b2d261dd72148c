"""Binary morphology with disk footprints, restated on SciPy: the three functions of scikit-image the reference's downstream
stages use (superdsm/postprocess.py:155,258-262,270,321; superdsm/render.py:380-384).  scikit-image is not a dependency here;
what these do is its documented behaviour: ``disk(r)`` = the pixels with dy^2 + dx^2 <= r^2 of a (2r+1)^2 window;
``binary_dilation`` treats everything outside the image as background, ``binary_erosion`` as foreground (the image border does
not erode)."""
import numpy as np
import scipy.ndimage as ndi


def disk(radius):
    r = int(radius)
    d = np.arange(-r, r + 1)
    return ((d[:, None] ** 2 + d[None, :] ** 2) <= r * r).astype(np.uint8)


def binary_dilation(image, footprint):
    return ndi.binary_dilation(np.asarray(image, bool), structure=np.asarray(footprint, bool))


def binary_erosion(image, footprint):
    return ndi.binary_erosion(np.asarray(image, bool), structure=np.asarray(footprint, bool), border_value=True)
