"""Candidate objects and their batched evaluation on the GPU.

Mirrors the reference's ``superdsm.objects`` API (superdsm/objects.py:12-284): ``Object`` carries a footprint (set of
atom labels) and, after :func:`compute_objects`, ``energy``, ``on_boundary``, ``is_optimal``, ``processing_time``,
``fg_offset`` and ``fg_fragment``.  ``compute_objects`` keeps the reference's signature and its in-place mutation
contract, but instead of one Ray task per candidate (objects.py:275-281) the whole list becomes ONE batch of the HIP
engine (region crops, G~ rows, elliptical + DSM solves, masks).  There is no CPU path.
"""
import os
import time
import warnings
import zlib

import numpy as np

from . import _capi
from .output import get_output

try:
    from xxhash import xxh3_64_intdigest as _xxh3
except ImportError:                                    # pragma: no cover
    _xxh3 = None


class BaseObject:
    """A segmentation mask stored as a minimal-size fragment plus its offset (objects.py:12-50)."""

    def __init__(self):
        self.fg_offset = None
        self._fg_fragment = None

    @property
    def fg_fragment(self):
        """The fragment (bool array).  compute_objects stores ``(engine.PackedFragments, index)`` -- the bit-packed masks of the batch --
        and the fragment is unpacked on first access: most candidates of a generation are pruned without their mask ever being looked at."""
        f = self._fg_fragment
        if type(f) is tuple:
            src, i = f
            f = self._fg_fragment = src.get(i)
        return f

    @fg_fragment.setter
    def fg_fragment(self, value):
        self._fg_fragment = value

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

# Batches of fewer candidates than this leave most compute units idle: they run in latency mode (256 threads per candidate, regions of
# more than 3072 pixels split over workgroup groups) also when they span several images (lock-step generations of a small image set).
# (1024 until round 3: a lock-step batch of 588 candidates of 4 images then took 23.5 ms -- hundreds of group members, a compute unit
# each -- against 11.6 ms in throughput mode; 8 different BBBC039-like images: 14.4-15.7 -> 13.3-14.2 ms per image with 256.)
LATENCY_MODE_BELOW = int(os.environ.get('SDSM_LATENCY_BELOW', 256))

# keys of dsm_cfg that only steer the reference's CPU implementation
_CPU_ONLY_KEYS = ('smooth_mat_max_allocations', 'cachesize', 'cachetest', 'smooth_mat_dtype', 'cp_timeout')


def _fingerprint(a):
    """Content fingerprint of an array: identity, shape and a hash of ALL of its bytes (xxh3: 0.1 ms for a 520 x 696 float64
    image; CRC-32 where xxhash is missing: 2.6 ms), so that an in-place edit of any pixel between two calls invalidates the
    cached device image."""
    a = np.ascontiguousarray(a)
    raw = a.view(np.uint8).reshape(-1)
    return (id(a), a.shape, str(a.dtype), _xxh3(raw) if _xxh3 is not None else zlib.crc32(raw))


def device_image(y, atoms, background_margin, refresh=False):
    """The per-image device state, cached on the ``Image`` object: y, atoms and the candidate-independent validity mask are
    uploaded / computed once, not once per candidate (objects.py:126-127 does the latter).  The cache is keyed by a content
    fingerprint (full hash) of ``atoms`` / ``y.model`` / ``y.mask`` and the margin; ``refresh=True`` rebuilds it unconditionally.
    ``y._sdsm_pinned = atoms`` (set by the stage on the Image it created itself, for the batches of one ``process`` call) skips the
    hashing -- 0.3 ms per batch and image."""
    from . import engine
    cache = getattr(y, '_sdsm_device', None)
    if cache is not None and not refresh and getattr(y, '_sdsm_pinned', None) is atoms and cache[0][3] == float(background_margin):
        return cache[1]                                     # pinned by the stage for the duration of one `process` (its own Image, same atoms array)
    key = (_fingerprint(atoms), _fingerprint(y.model), None if y.mask is None else _fingerprint(y.mask), float(background_margin))
    if refresh or cache is None or cache[0] != key:
        mask = None if y.mask is None or y.mask.all() else y.mask
        cache = (key, engine.DeviceImage(y.model, mask, atoms, background_margin))
        y._sdsm_device = cache
    if hasattr(y, '_sdsm_pinned') and y._sdsm_pinned is None:
        y._sdsm_pinned = atoms                              # the stage opted in (GlobalEnergyMinimization.process): no hashing for the rest of the call
    return cache[1]


_device_image = device_image


def compute_norm_energy(obj):
    """``postprocess._compute_norm_energy`` (postprocess.py:289-291) without recomputing the region: compute_objects keeps its size."""
    return obj.energy / obj.cvxprog_region_size


_STATUS_NAMES = {_capi.CAND_OPTIMAL: 'optimal', _capi.CAND_FALLBACK: 'fallback', _capi.CAND_TRIVIAL: 'trivial (single positive pixel)',
                 _capi.CAND_ERROR: 'error', _capi.CAND_UNSUPPORTED: 'unsupported', _capi.CAND_GIVEN_UP: 'given up (scheduling)'}


def _write_logs(log_root_dir, records):
    """One ``<cidx>.txt`` per candidate, as the reference's workers do (objects.py:220-237 redirect the solver's output there);
    here the solver's counters and outcome."""
    if log_root_dir is None:
        return
    os.makedirs(log_root_dir, exist_ok=True)
    for cidx, r in enumerate(records):
        with open(os.path.join(log_root_dir, f'{cidx}.txt'), 'w') as f:
            f.write(f'status: {_STATUS_NAMES.get(int(r["status"]), int(r["status"]))}\n'
                    f'region pixels N: {int(r["n_pixels"])}  deformation parameters M: {int(r["n_deform"])}\n'
                    f'elliptical model: {int(r["iters_ell"])} Newton iterations, psi = {float(r["energy_ell"])!r}'
                    f'{" (retry from the moment initialisation)" if int(r["flags"]) & 1 else ""}\n'
                    f'deformable model: {int(r["iters_dsm"])} Newton iterations, psi = {float(r["energy"])!r}\n'
                    f'passes over the pixels: {int(r["evals_full"])} full, {int(r["evals_value"])} value / line search\n'
                    f'theta: {np.asarray(r["theta"]).tolist()}\n')


def _solve(images, footprints, image_of, cfg, shard, while_waiting=None):
    """One batch over one or several images -> (records, fragments).  Candidates whose workgroup group was given up (a
    scheduling event on an oversubscribed GPU, not a solver failure) are solved again without groups."""
    from . import engine
    import torch
    if shard is not None:
        assert len(images) == 1, 'sharded batches cover one image'
        return shard.solve(images[0], footprints, cfg)
    batch = engine.Batch(images if len(images) > 1 else images[0], footprints, cfg, image_of=image_of,
                         mode=1 if len(footprints) < LATENCY_MODE_BELOW else 0)   # a batch that cannot fill the GPU: shortest wall clock
    start = _starting_points(batch, cfg)
    batch.launch()
    if while_waiting is not None:                          # host work of the caller while the kernels run (the launch is asynchronous)
        while_waiting()
    records, masks = batch.download()
    records = records.copy()
    fragments = batch.fragments(records, masks=masks, lazy=True)
    again = np.flatnonzero(records['status'] == _capi.CAND_GIVEN_UP)
    if again.size:
        sub = engine.Batch(images if len(images) > 1 else images[0], [footprints[i] for i in again], cfg,
                           image_of=None if image_of is None else np.asarray(image_of)[again], mode=2)
        if start is not None:
            sub.set_start([start[i] for i in again])
        sub.launch()
        rec2, masks2 = sub.download()
        frag2 = sub.fragments(rec2, masks=masks2, lazy=True)
        for j, i in enumerate(again):
            records[i] = rec2[j]
            fragments[i] = frag2[j]
    return records, fragments


def _starting_points(batch, cfg):
    """Callable ``dsm/init`` (objects.py:385-386): ``params = init(number of columns of G~)`` per candidate -- the count is a result of
    the setup kernel, which runs once on its own for it -- handed to the batch as the starting points of its DSM solves.  Returns the
    list of vectors (None for candidates without a solve), or None if ``init`` is not callable."""
    init = cfg.get('init')
    if not callable(init):
        return None
    start = []
    for i, m in enumerate(batch.deform_counts().tolist()):
        if m < 0:
            start.append(None)
            continue
        p = np.asarray(init(m), np.float64).ravel()
        if p.size != 6 + m or not np.all(np.isfinite(p)):
            raise ValueError(f'dsm/init({m}) must return {6 + m} finite parameters (candidate {i}: got {p.size})')
        start.append(p)
    batch.set_start(start)
    return start


def _assign(objects, records, fragments, dt, cidx0=0):
    fallbacks = 0
    status = records['status'].tolist()
    energy = records['energy'].tolist()
    onb = records['on_boundary'].tolist()
    npx = records['n_pixels'].tolist()
    per = dt / max(1, len(objects))
    for k, obj in enumerate(objects):
        st = status[k]
        if st == _capi.CAND_ERROR:
            raise CvxprogError('convex programming failed for the elliptical model', cidx=cidx0 + k)
        if st == _capi.CAND_UNSUPPORTED:
            if int(records['evals_full'][k]) == 0:           # rejected by the setup kernel (bounding box / grid beyond its tables): no result at all
                raise _capi.SdsmError(f'candidate {cidx0 + k} exceeds an implementation limit of the GPU solver '
                                      f'(N={npx[k]}, M={int(records["n_deform"][k])}); see DESIGN.md "Limits"')
            # more deformation parameters than the solve kernels hold: the elliptical solution is the result, as after a failed DSM
            # solve in the reference (objects.py:399-410: failure => fallback, never an abort of the batch)
            warnings.warn(f'candidate {cidx0 + k}: 6 + M = {6 + int(records["n_deform"][k])} parameters exceed the GPU solver\'s limit; '
                          f'the elliptical solution is returned (is_optimal = False), see DESIGN.md "Limits"', RuntimeWarning, stacklevel=3)
        if st == _capi.CAND_GIVEN_UP:
            raise _capi.SdsmError(f'candidate {cidx0 + k}: the GPU could not schedule its workgroup group, neither on the attempt without groups (not a solver failure)')
        obj.fg_offset, obj.fg_fragment = fragments[k]
        obj.energy = energy[k]
        obj.on_boundary = bool(onb[k])
        obj.is_optimal = st == _capi.CAND_OPTIMAL
        # size of the convex-programming region: what postprocess._compute_norm_energy divides by (postprocess.py:289-291; the
        # reference recomputes the region with a full-image distance transform per object)
        obj.cvxprog_region_size = npx[k]
        # the batch is solved concurrently: the wall time is attributed evenly (the reference records the per-task time)
        obj.processing_time = 0 if st == _capi.CAND_TRIVIAL else per
        fallbacks += st in (_capi.CAND_FALLBACK, _capi.CAND_UNSUPPORTED)
    return fallbacks


def _clean_cfg(dsm_cfg):
    cfg = {k: v for k, v in dsm_cfg.items() if k not in _CPU_ONLY_KEYS}
    # dsm/hessian_sparsity_tol (dsm.py:377-383) only drops small entries of the Hessian that the reference hands to its solver: psi and its gradient -- hence the optimum -- do not
    # change, and the solver here uses an approximate Hessian of its own: accepted and not needed.  dsm/sparsity_tol also zeroes small residuals in the
    # GRADIENT (dsm.py:346) and curvature weights (dsm.py:362): it moves the point the reference's solver stops at, in a way only cvxopt's iterates define -- refused.
    hst = cfg.pop('hessian_sparsity_tol', 0)
    if not (hst >= 0):
        raise AssertionError('hessian_sparsity_tol must be positive')                  # dsm.py:285
    if cfg.pop('sparsity_tol', 0) != 0:
        raise NotImplementedError('dsm/sparsity_tol != 0 changes the reference\'s results (dsm.py:346,362) and is not implemented by the GPU solver')
    return cfg


def compute_objects(objects, y, atoms, dsm_cfg, log_root_dir, status_line=DEFAULT_COMPUTING_STATUS_LINE, out=None, shard=None, while_waiting=None):
    """Computes ``energy``, ``on_boundary``, ``is_optimal``, ``processing_time``, ``fg_offset`` and ``fg_fragment`` of
    every object IN PLACE (objects.py:243-267).

    :param objects: iterable of :class:`Object` (only ``footprint`` is read).
    :param y: :class:`~superdsm_amd.image.Image` of offset intensities (``model``) with its ``mask``.
    :param atoms: int image of atom labels.
    :param dsm_cfg: the ``dsm/*`` hyper-parameters (dsmcfg.py:6-21).
    :param log_root_dir: directory that receives one ``<cidx>.txt`` per candidate (objects.py:220-237), or ``None``.
    :param shard: optional :class:`superdsm_amd.dist.Sharder`: solve only this rank's share, all-gather the results.
    :param while_waiting: optional callable (extension): host work of the caller, run once after the batch has been launched and before
        its results are waited for (the launch is asynchronous); exceptions it raises propagate.
    """
    out = get_output(out)
    objects = list(objects)
    cfg = _clean_cfg(dsm_cfg)
    if len(objects) == 0:
        if while_waiting is not None:
            while_waiting()
        out.write(f'{status_line[1]}: 0 (0x fallback)')
        return
    margin = cfg.pop('background_margin', 20)
    image = device_image(y, atoms, margin)
    out.intermediate(f'{status_line[0]}... 0 / {len(objects)}')
    t0 = time.time()
    records, fragments = _solve([image], [obj.footprint for obj in objects], None, cfg, shard, while_waiting if shard is None else None)
    if shard is not None and while_waiting is not None:
        while_waiting()
    dt = time.time() - t0
    _write_logs(log_root_dir, records)
    fallbacks = _assign(objects, records, fragments, dt)
    out.write(f'{status_line[1]}: {len(objects)} ({fallbacks}x fallback)')


def compute_objects_multi(jobs, dsm_cfg, log_root_dirs=None, status_line=DEFAULT_COMPUTING_STATUS_LINE, out=None, while_waiting=None, errors=None):
    """:func:`compute_objects` for several images at once: ``jobs`` is a list of ``(objects, y, atoms)``, all solved as ONE
    batch of the engine (sdsm_plan_create_multi) -- the batches of a single small image (tens of candidates per generation,
    globalenergymin.py:357) cannot fill a GPU, the same generation of several images can.  Results are set in place, job by job;
    ``cidx`` of an error counts within its job.  More than 16 images are processed in groups of 16.  ``errors``: a list (one slot per
    job) that receives the exception of a job whose candidate failed (CvxprogError, an implementation limit) instead of raising it for
    the whole batch: the other images' results are complete -- the reference fails one image, not the set."""
    out = get_output(out)
    cfg = _clean_cfg(dsm_cfg)
    margin = cfg.pop('background_margin', 20)
    jobs = [(list(objs), y, atoms) for objs, y, atoms in jobs]
    todo = [j for j, (objs, _, _) in enumerate(jobs) if len(objs)]
    total = fb = 0
    per_job = [0] * len(jobs)                               # fallbacks of every job (returned)
    for g0 in range(0, len(todo), 16):
        group = todo[g0:g0 + 16]
        images = [device_image(jobs[j][1], jobs[j][2], margin) for j in group]
        fps, image_of = [], []
        for k, j in enumerate(group):
            fps.extend(obj.footprint for obj in jobs[j][0])
            image_of.extend([k] * len(jobs[j][0]))
        t0 = time.time()
        records, fragments = _solve(images, fps, np.asarray(image_of, np.int32), cfg, None, while_waiting)
        while_waiting = None                                 # (once: while the first batch is on the GPU)
        dt = time.time() - t0
        pos = 0
        for j in group:
            objs = jobs[j][0]
            rec, frs = records[pos:pos + len(objs)], fragments[pos:pos + len(objs)]
            if log_root_dirs is not None and log_root_dirs[j] is not None:
                _write_logs(log_root_dirs[j], rec)
            try:
                per_job[j] = _assign(objs, rec, frs, dt * len(objs) / len(fps))
            except (CvxprogError, _capi.SdsmError) as error:
                if errors is None:
                    raise
                errors[j] = error
            fb += per_job[j]
            pos += len(objs)
            total += len(objs)
    if while_waiting is not None:                            # (no batch at all: nothing to wait for)
        while_waiting()
    out.write(f'{status_line[1]}: {total} ({fb}x fallback)')
    return per_job
