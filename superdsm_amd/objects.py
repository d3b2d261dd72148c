"""Candidate objects and their batched evaluation on the GPU.

Mirrors the reference's ``superdsm.objects`` API (superdsm/objects.py:12-284): ``Object`` carries a footprint (set of
atom labels) and, after :func:`compute_objects`, ``energy``, ``on_boundary``, ``is_optimal``, ``processing_time``,
``fg_offset`` and ``fg_fragment``.  ``compute_objects`` keeps the reference's signature and its in-place mutation
contract, but instead of one Ray task per candidate (objects.py:275-281) the whole list becomes ONE batch of the HIP
engine (region crops, G~ rows, elliptical + DSM solves, masks).  There is no CPU path.
"""
import time

import numpy as np

from . import _capi
from .output import get_output


class BaseObject:
    """A segmentation mask stored as a minimal-size fragment plus its offset (objects.py:12-50)."""

    def __init__(self):
        self.fg_offset = None
        self.fg_fragment = None

    def fill_foreground(self, out, value=True):
        assert self.fg_offset is not None and self.fg_fragment is not None
        h, w = self.fg_fragment.shape
        r, c = int(self.fg_offset[0]), int(self.fg_offset[1])
        sel = np.s_[r:r + h, c:c + w]
        out[sel] = value * self.fg_fragment
        return sel


class Object(BaseObject):
    """A set of atomic image regions and the result of its convex energy minimisation (objects.py:53-145)."""

    def __init__(self):
        super().__init__()
        self.footprint = set()
        self.energy = np.nan
        self.on_boundary = np.nan
        self.is_optimal = np.nan
        self.processing_time = np.nan
        self.cvxprog_region_size = 0      # extension: pixels of the convex-programming region (compute_norm_energy)

    def get_mask(self, atoms):
        return np.isin(atoms, list(self.footprint))

    def set(self, state):
        self.fg_fragment = None if state.fg_fragment is None else state.fg_fragment.copy()
        self.fg_offset = None if state.fg_offset is None else state.fg_offset.copy()
        self.footprint = set(state.footprint)
        for attr in ('energy', 'on_boundary', 'is_optimal', 'processing_time'):
            setattr(self, attr, getattr(state, attr))
        self.cvxprog_region_size = getattr(state, 'cvxprog_region_size', 0)
        return self

    def copy(self):
        return Object().set(self)


def extract_foreground_fragment(fg_mask):
    """Minimal bounding rectangle of a mask and its offset; an empty mask gives ``(0, 0)`` and ``[[False]]``
    (objects.py:148-174)."""
    rows = np.flatnonzero(fg_mask.any(axis=1))
    if rows.size == 0:
        return np.zeros(2, int), np.zeros((1, 1), bool)
    cols = np.flatnonzero(fg_mask.any(axis=0))
    return np.array([rows[0], cols[0]]), fg_mask[rows[0]:rows[-1] + 1, cols[0]:cols[-1] + 1]


class CvxprogError(Exception):
    """The elliptical solve failed twice without producing any solution (objects.py:309-318, 351-353)."""

    def __init__(self, *args, cidx=None):
        super().__init__(*args)
        self.cidx = cidx

    def __str__(self):
        parts = [str(a) for a in self.args]
        if self.cidx is not None:
            parts.append(f'cidx: {self.cidx}')
        return ', '.join(parts)


DEFAULT_COMPUTING_STATUS_LINE = ('Computing objects', 'Computed objects')

# keys of dsm_cfg that only steer the reference's CPU implementation
_CPU_ONLY_KEYS = ('smooth_mat_max_allocations', 'cachesize', 'cachetest', 'smooth_mat_dtype', 'cp_timeout')


def _device_image(y, atoms, background_margin):
    """The per-image device state is cached on the ``Image`` object: y, atoms and the candidate-independent
    validity mask are uploaded / computed once, not once per candidate (objects.py:126-127 does the latter)."""
    from . import engine
    cache = getattr(y, '_sdsm_device', None)
    key = (id(atoms), float(background_margin))
    if cache is None or cache[0] != key:
        mask = None if y.mask is None or y.mask.all() else y.mask
        cache = (key, engine.DeviceImage(y.model, mask, atoms, background_margin))
        y._sdsm_device = cache
    return cache[1]


def compute_norm_energy(obj):
    """``postprocess._compute_norm_energy`` (postprocess.py:289-291) without recomputing the region: compute_objects keeps its size."""
    return obj.energy / obj.cvxprog_region_size


def compute_objects(objects, y, atoms, dsm_cfg, log_root_dir, status_line=DEFAULT_COMPUTING_STATUS_LINE, out=None, shard=None):
    """Computes ``energy``, ``on_boundary``, ``is_optimal``, ``processing_time``, ``fg_offset`` and ``fg_fragment`` of
    every object IN PLACE (objects.py:243-267).

    :param objects: iterable of :class:`Object` (only ``footprint`` is read).
    :param y: :class:`~superdsm_amd.image.Image` of offset intensities (``model``) with its ``mask``.
    :param atoms: int image of atom labels.
    :param dsm_cfg: the ``dsm/*`` hyper-parameters (dsmcfg.py:6-21).
    :param log_root_dir: accepted for API compatibility; per-candidate log files are not written.
    :param shard: optional :class:`superdsm_amd.dist.Sharder`: solve only this rank's share, all-gather the results.
    """
    from . import engine
    import torch
    out = get_output(out)
    objects = list(objects)
    cfg = {k: v for k, v in dsm_cfg.items() if k not in _CPU_ONLY_KEYS}
    if callable(cfg.get('init')):
        raise NotImplementedError('dsm/init as a callable is not supported by the GPU solver')
    if len(objects) == 0:
        out.write(f'{status_line[1]}: 0 (0x fallback)')
        return
    margin = cfg.pop('background_margin', 20)
    image = _device_image(y, atoms, margin)
    out.intermediate(f'{status_line[0]}... 0 / {len(objects)}')
    t0 = time.time()
    footprints = [sorted(int(a) for a in obj.footprint) for obj in objects]
    if shard is not None:
        records, fragments = shard.solve(image, footprints, cfg)
    else:
        batch = engine.Batch(image, footprints, cfg, latency_mode=True)     # one image at a time: shortest wall clock
        batch.launch()
        torch.cuda.synchronize(image.device)
        records = batch.records()
        fragments = batch.fragments(records)
    dt = time.time() - t0
    fallbacks = 0
    for cidx, (obj, rec, (off, frag)) in enumerate(zip(objects, records, fragments)):
        status = int(rec['status'])
        if status == _capi.CAND_ERROR:
            raise CvxprogError('convex programming failed for the elliptical model', cidx=cidx)
        if status == _capi.CAND_UNSUPPORTED:
            raise _capi.SdsmError(f'candidate {cidx} exceeds an implementation limit of the GPU solver '
                                  f'(N={int(rec["n_pixels"])}, M={int(rec["n_deform"])}); see DESIGN.md "Limits"')
        obj.fg_offset, obj.fg_fragment = off, frag
        obj.energy = float(rec['energy'])
        obj.on_boundary = bool(rec['on_boundary'])
        obj.is_optimal = status == _capi.CAND_OPTIMAL
        # size of the convex-programming region: what postprocess._compute_norm_energy divides by (postprocess.py:289-291; the
        # reference recomputes the region with a full-image distance transform per object)
        obj.cvxprog_region_size = int(rec['n_pixels'])
        # the batch is solved concurrently: the wall time is attributed evenly (the reference records the per-task time)
        obj.processing_time = 0 if status == _capi.CAND_TRIVIAL else dt / len(objects)
        fallbacks += status == _capi.CAND_FALLBACK
    out.write(f'{status_line[1]}: {len(objects)} ({fallbacks}x fallback)')
