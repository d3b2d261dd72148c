"""Import stand-ins that let /root/reference's ``superdsm`` package be imported unmodified in the
build container: the third-party modules it imports at module level are absent
here (ray, cvxopt, cvxpy, scikit-image, IPython).  None of the stand-ins computes anything on the
solver's arithmetic path:

* ``ray`` / ``cvxopt`` / ``cvxpy`` / ``IPython``: empty modules (never called by the fixture generator);
* ``skimage.util.view_as_windows``: ``numpy.lib.stride_tricks.sliding_window_view`` (a strided view,
  no arithmetic), used by the reference's ``_convmat`` (superdsm/dsm.py:156);
* ``skimage.morphology.disk(1)`` / ``binary_dilation``: the 4-neighbourhood cross and SciPy's binary
  dilation, used only by ``AtomAdjacencyGraph.__init__`` (superdsm/atoms.py:62-75) for the
  adjacency fixtures;
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
    cross = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)
    sk.morphology = _mod('skimage.morphology',
                         disk=lambda r: cross if r == 1 else (_ for _ in ()).throw(NotImplementedError()),
                         binary_dilation=lambda img, se: scipy.ndimage.binary_dilation(img, structure=se.astype(bool)))
    for sub in ('segmentation', 'io', 'feature', 'measure', 'filters', 'draw', 'color'):
        setattr(sk, sub, _mod('skimage.' + sub))
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
