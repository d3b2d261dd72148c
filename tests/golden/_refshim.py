"""Import stand-ins that let /root/reference's ``superdsm`` package be imported unmodified in the
build container: the third-party modules it imports at module level are absent
here (ray, cvxopt, cvxpy, scikit-image, IPython).  None of the stand-ins computes anything on the
solver's arithmetic path:

* ``ray`` / ``cvxopt`` / ``cvxpy`` / ``IPython``: empty modules (never called by the fixture generator);
* ``skimage.util.view_as_windows``: ``numpy.lib.stride_tricks.sliding_window_view`` (a strided view,
  no arithmetic), used by the reference's ``_convmat`` (superdsm/dsm.py:156);
* ``skimage.morphology.disk`` / ``binary_dilation`` / ``binary_erosion``: their documented behaviour on SciPy
  (disk = dy^2 + dx^2 <= r^2; dilation with background outside the image, erosion with foreground outside), used
  by ``AtomAdjacencyGraph.__init__`` (superdsm/atoms.py:62-75) for the adjacency fixtures and by the
  post-processing / rasterisation fixtures (superdsm/postprocess.py:155,270,321; superdsm/render.py:380-384);
  ``skimage.segmentation.watershed``: exact only when there is nothing to flood (every masked pixel is a marker or
  cannot be reached from one), refuses otherwise -- the render fixtures contain only such cases;
* a ``.A`` property on SciPy sparse classes (removed in SciPy >= 1.14, used by the vendored MKL wrapper).

This module only runs in the container that holds /root/reference; it never travels to the GPU box
as anything but text, and nothing imports it there.
"""
import os
import sys
import types

import numpy as np
import scipy.ndimage
import scipy.sparse
from numpy.lib.stride_tricks import sliding_window_view

REFERENCE_ROOT = '/root/reference'


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if 'superdsm' in sys.modules:
        return
    _mod('ray', remote=lambda f: f, put=lambda x: x, get=lambda x: x, wait=None, init=lambda **kw: None)
    _mod('cvxopt', matrix=None, spmatrix=None, solvers=None)
    _mod('cvxpy')
    sk = _mod('skimage')
    sk.util = _mod('skimage.util', view_as_windows=lambda arr, shape: sliding_window_view(arr, shape))
    def disk(r):                      # skimage.morphology.disk: the pixels with dy^2 + dx^2 <= r^2 of a (2r+1)^2 window
        d = np.arange(-int(r), int(r) + 1)
        return ((d[:, None] ** 2 + d[None, :] ** 2) <= int(r) ** 2).astype(np.uint8)
    sk.morphology = _mod('skimage.morphology', disk=disk,
                         binary_dilation=lambda img, se: scipy.ndimage.binary_dilation(img, structure=se.astype(bool)),
                         binary_erosion=lambda img, se: scipy.ndimage.binary_erosion(img, structure=se.astype(bool), border_value=True))
    for sub in ('segmentation', 'io', 'feature', 'measure', 'filters', 'draw', 'color'):
        setattr(sk, sub, _mod('skimage.' + sub))

    def watershed_nothing_to_flood(image, markers, mask=None):
        # stand-in for skimage.segmentation.watershed that is exact whenever every masked pixel already carries a marker (no
        # flooding takes place) and refuses anything else: the render fixtures only contain such cases
        out = np.where(mask, markers, 0) if mask is not None else markers.copy()
        if mask is not None and (mask & (out == 0)).any():
            # unlabelled masked pixels are fine if no marker can reach them (their 4-connected component of the mask holds no
            # marker: they stay 0, render.py:443-447 deals with them); anything else would need real flooding
            comp, n = scipy.ndimage.label(mask)
            for l in np.unique(comp[mask & (out == 0)]):
                if (out[comp == l] > 0).any():
                    raise NotImplementedError('watershed flooding is scikit-image arithmetic')
        return out.astype(np.int32)
    sk.segmentation.watershed = watershed_nothing_to_flood
    tr = _mod('skimage.transform')
    tr.__path__ = []
    sk.transform = tr
    _mod('skimage.transform._warps')
    _mod('IPython')
    _mod('IPython.display', clear_output=None, display=None)
    _mod('matplotlib')
    _mod('matplotlib.pyplot')
    for cls in (scipy.sparse.csr_matrix, scipy.sparse.csc_matrix, scipy.sparse.coo_matrix):
        if not hasattr(cls, 'A'):
            cls.A = property(lambda self: self.toarray())
    os.environ.setdefault('MKL_NUM_THREADS', '2')
    sys.path.insert(0, REFERENCE_ROOT)
