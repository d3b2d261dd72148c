"""``global-energy-minimization`` stage: Algorithm 1 + Criterion 2 of Kostrykin & Rohr (TPAMI 2023) around the
batched GPU candidate solver.

Drop-in for the reference stage (superdsm/globalenergymin.py:97-368): same stage name, inputs
(``y, y_mask, atoms, adjacencies, dsm_cfg``), outputs (``y_img, cover, objects, performance``), hyper-parameters
(``pruning, beta, max_iter, gamma, max_seed_distance, max_work_amount``) and ``configure_ex`` factors.  The
generation / pruning logic is host-side bookkeeping and is restated here; every ``compute_objects`` call is one GPU
batch.
"""
import gc
import math
import os
import sys
import threading
import time

import numpy as np

from .image import Image
from .maxsetpack import solve_maxsetpack
from .minsetcover import DEFAULT_GAMMA, DEFAULT_MAX_ITER, MinSetCover
from .objects import DEFAULT_COMPUTING_STATUS_LINE, CvxprogError, Object, compute_objects, compute_objects_multi
from .output import Text, get_output
from .pipeline import Stage

DEFAULT_MAX_WORK_AMOUNT = 10 ** 6
DEFAULT_SPECULATION = 8            # generations solved ahead per GPU batch (extension; 0 = the reference's batches exactly)
DEFAULT_SPECULATION_BUDGET = 2048  # ... while the batch stays within this many candidates (measured, BBBC039-like images, depth / budget: 3 / 768 = 2-3 batches
                                   # per image, 9.7 / 17.2 / 35.3 ms on three layouts; 8 / 2048 = 1-2 batches, 9.1 / 14.5 / 32.8 ms; candidates solved in vain: the same 32-120)
DEFAULT_LOCKSTEP_BUDGET = 4096     # the same for the multi-image batches of process_many, all images together


class PerformanceReport:
    """Pruning statistics (globalenergymin.py:23-94)."""

    attributes = [
        'direct_solution_trial_count',
        'direct_solution_success_count',
        'iterative_object_count',
        'iterative_computed_object_count',
        'overall_object_count',
        'overall_computed_object_count',
        'nontrivial_object_count',
        'nontrivial_computed_object_count',
    ]

    def __init__(self, **kwargs):
        for key in self.attributes:
            setattr(self, key, kwargs.get(key, 0))

    @staticmethod
    def _ratio(num, den, complement=False):
        if den == 0:
            return np.nan
        return 1 - num / den if complement else num / den

    @property
    def direct_solution_success(self):
        return self._ratio(self.direct_solution_success_count, self.direct_solution_trial_count)

    @property
    def iterative_pruning_success(self):
        return self._ratio(self.iterative_computed_object_count, self.iterative_object_count, True)

    @property
    def overall_pruning_success(self):
        return self._ratio(self.overall_computed_object_count, self.overall_object_count, True)

    @property
    def nontrivial_pruning_success(self):
        return self._ratio(self.nontrivial_computed_object_count, self.nontrivial_object_count, True)

    def __iadd__(self, other):
        for key in self.attributes:
            setattr(self, key, getattr(self, key) + getattr(other, key))
        return self

    def _assert_integrity(self):
        for value in (self.direct_solution_success, self.iterative_pruning_success, self.nontrivial_pruning_success, self.overall_pruning_success):
            assert np.isnan(value) or 0 <= value <= 1, value


def _generation_log_dir(log_root_dir, generation_number):
    if log_root_dir is None:
        return None
    path = os.path.join(log_root_dir, f'gen{generation_number}')
    os.makedirs(path, exist_ok=True)
    return path


def _within_seed_distance(footprint, new_atom, adjacencies, max_seed_distance):
    if math.isinf(max_seed_distance):
        return True
    seed = np.asarray(adjacencies.get_seed(new_atom))
    return all(np.linalg.norm(np.asarray(adjacencies.get_seed(a)) - seed) <= max_seed_distance for a in footprint)


def _expand(footprint, adjacencies, max_seed_distance, skip_last):
    """``(cluster, ((grown footprint, new atom), ...))``: the footprint grown by every adjacent atom within the seed distance."""
    cluster = adjacencies.get_cluster_label(next(iter(footprint)))
    if skip_last and len(footprint) + 1 == len(adjacencies.get_atoms_in_cluster(cluster)):
        return cluster, ()                                  # the universe is computed separately
    neighbours = set()
    for atom in footprint:
        neighbours |= adjacencies[atom] - footprint
    if math.isinf(max_seed_distance):                        # (math, not np.isinf -- 0.6 us on a scalar --, and a list instead of a generator: a third of this function)
        return cluster, tuple([(frozenset(footprint | {new_atom}), new_atom) for new_atom in neighbours])
    return cluster, tuple([(frozenset(footprint | {new_atom}), new_atom) for new_atom in neighbours
                           if _within_seed_distance(footprint, new_atom, adjacencies, max_seed_distance)])


def _iterate_generation(previous_generation, adjacencies, max_seed_distance, get_footprint=lambda item: item,
                        ignored_cluster_labels=frozenset(), skip_last=False, memo=None):
    """Yields ``(item, new_footprint, new_atom)``: every footprint of the previous generation grown by one adjacent
    atom, de-duplicated within the generation (globalenergymin.py:292-307).  ``memo``: a dict that keeps the expansions of the
    footprints across calls (same adjacencies, seed distance and ``skip_last``): the stage enumerates what is left of the search
    space for its statistics, then the generations themselves, then the candidates to solve ahead -- the same footprints."""
    seen = set()
    for item in previous_generation:
        footprint = get_footprint(item)
        if memo is None:
            cluster, children = _expand(footprint, adjacencies, max_seed_distance, skip_last)
        else:
            key = footprint if type(footprint) is frozenset else frozenset(footprint)
            entry = memo.get(key)
            if entry is None:
                entry = memo[key] = _expand(footprint, adjacencies, max_seed_distance, skip_last)
            cluster, children = entry
        if cluster in ignored_cluster_labels:
            continue
        for grown, new_atom in children:
            if grown not in seen:
                seen.add(grown)
                yield item, grown, new_atom


def _remaining_by_cluster(generations, adjacencies, max_seed_distance, max_amount=DEFAULT_MAX_WORK_AMOUNT, skip_last=False, memo=None):
    """Footprints still to be enumerated after the last generation, per cluster (a footprint never leaves its cluster).  From the
    generation of the atoms on (the only way the stage calls it) the counting is native host code (sdsm_count_growth: bit sets per
    cluster, same numbers -- tested against the enumeration below); clusters of more than 64 atoms are enumerated here."""
    current = [c.footprint for c in generations[-1]]
    remaining = {}
    total = 0
    if len(generations) == 1 and all(len(fp) == 1 for fp in current):
        counted = _count_growth_native(adjacencies, {next(iter(fp)) for fp in current}, max_seed_distance, skip_last, max_amount)
        if counted is not None:
            counts, current = counted                       # current: the atoms of the clusters that were too large for the bit sets
            for cl, n in counts.items():
                if n > 0:
                    remaining[cl] = n
                    total += n
            if total > max_amount:
                raise ValueError('estimated work amount is too large')
    while current:
        current = [fp for _, fp, _ in _iterate_generation(current, adjacencies, max_seed_distance, skip_last=skip_last, memo=memo)]
        for fp in current:
            cl = adjacencies.get_cluster_label(next(iter(fp)))
            remaining[cl] = remaining.get(cl, 0) + 1
        total += len(current)
        if total > max_amount:                              # (what the first of the reference's two enumerations would see: clusters of <= 2 atoms have nothing to enumerate)
            raise ValueError('estimated work amount is too large')
    return remaining


def _count_growth_native(adjacencies, atom_labels, max_seed_distance, skip_last, max_amount):
    """``({cluster: count}, [footprints of the atoms of clusters left to the caller])`` or None if the native library is missing."""
    try:
        from . import _capi
        L = _capi.lib()
    except Exception:                                       # noqa: BLE001 -- host logic must work without the library (CPU tests of the logic)
        return None
    clusters = [cl for cl in adjacencies.cluster_labels]
    offsets = np.zeros(len(clusters) + 1, np.int32)
    adj, compat, members = [], [], []
    bounded = not np.isinf(max_seed_distance)
    for k, cl in enumerate(clusters):
        atoms = [a for a in sorted(adjacencies.get_atoms_in_cluster(cl)) if a in atom_labels]
        members.append(atoms)
        offsets[k + 1] = offsets[k] + len(atoms)
        if len(atoms) > 64:
            adj.extend([0] * len(atoms))
            compat.extend([0] * len(atoms))
            continue
        bit = {a: i for i, a in enumerate(atoms)}
        for a in atoms:
            m = 0
            for b in adjacencies[a]:
                if b in bit:
                    m |= 1 << bit[b]
            adj.append(m)
        if bounded:
            seeds = np.asarray([adjacencies.get_seed(a) for a in atoms], float).reshape(len(atoms), -1)
            for i in range(len(atoms)):
                # the reference's test, value by value: np.linalg.norm(seed_a - seed_new) <= max_seed_distance (globalenergymin.py:285-289)
                m = 0
                for j in range(len(atoms)):
                    if np.linalg.norm(seeds[j] - seeds[i]) <= max_seed_distance:
                        m |= 1 << j
                compat.append(m)
    adj_a = np.array(adj, np.uint64) if adj else np.zeros(1, np.uint64)
    compat_a = np.array(compat, np.uint64) if bounded and compat else None
    counts = np.zeros(max(1, len(clusters)), np.int64)
    p = lambda a: a.ctypes.data
    code = L.sdsm_count_growth(len(clusters), p(offsets), p(adj_a), p(compat_a) if compat_a is not None else None, int(bool(skip_last)),
                               int(max_amount), p(counts))
    assert code == 0, 'sdsm_count_growth: bad argument'
    rest = [frozenset([a]) for k in range(len(clusters)) if counts[k] < 0 for a in members[k]]
    return {cl: int(counts[k]) for k, cl in enumerate(clusters) if counts[k] >= 0}, rest


def _estimate_progress(generations, adjacencies, max_seed_distance, max_amount=DEFAULT_MAX_WORK_AMOUNT,
                       ignored_cluster_labels=frozenset(), skip_last=False, by_cluster=None, memo=None):
    """``(finished, remaining)`` (globalenergymin.py:310-323).  ``by_cluster``: the result of :func:`_remaining_by_cluster` for the
    same state (one enumeration serves several sets of ignored clusters)."""
    if by_cluster is None:
        current = [c.footprint for c in generations[-1]]
        remaining = 0
        while current:
            current = [fp for _, fp, _ in _iterate_generation(current, adjacencies, max_seed_distance,
                                                              ignored_cluster_labels=ignored_cluster_labels, skip_last=skip_last, memo=memo)]
            remaining += len(current)
            if remaining > max_amount:
                raise ValueError('estimated work amount is too large')
    else:
        remaining = sum(n for cl, n in by_cluster.items() if cl not in ignored_cluster_labels)
        if remaining > max_amount:
            raise ValueError('estimated work amount is too large')
    return sum(len(gen) for gen in generations), remaining


class _Speculation:
    """Wraps a solver with the signature of :func:`compute_objects`: every batch also solves the CHILDREN (``depth`` levels of
    :func:`_iterate_generation`) of the candidates it was asked for, and keeps them; a later batch is served from that store and
    only what is missing goes to the GPU -- nothing at all if everything is there.

    The generations of an image are sequential (the thresholds of generation k + 1 need the energies of generation k,
    globalenergymin.py:326-368) and hold tens of candidates each, while one batch of the engine takes as long as its largest
    region whether it has 20 or 500 candidates (a single image uses a fraction of an MI355X): solving the next generation
    for ALL parents, before it is known which of them survive the pruning, turns two dependent round trips into one.
    A batch is only extended while it stays within ``budget`` candidates -- what the GPU runs concurrently; beyond that the extra
    candidates would cost time instead of hiding latency (image sets in lock step share the budget).  The host
    logic is unchanged and sees the same candidates with the same results; candidates solved in vain are not counted in the
    PerformanceReport (``performance.speculative_object_count``)."""

    def __init__(self, solve, adjacencies, max_seed_distance, depth, ignored_cluster_labels, budget=DEFAULT_SPECULATION_BUDGET, memo=None):
        self.memo = memo
        self.solve, self.adjacencies, self.max_seed_distance, self.depth = solve, adjacencies, max_seed_distance, int(depth)
        self.budget = budget                                # largest batch that is still extended (candidates)
        self.ignored = ignored_cluster_labels               # a set the caller fills in (clusters solved directly)
        self.store = {}                                     # frozenset(footprint) -> solved Object
        self.batches = self.extra = self.served = 0         # GPU batches, candidates solved ahead, those of them asked for later
        self.unasked = set()                                # solved ahead, not asked for (yet)
        # host work to do while a batch is on the GPU (set by the caller): handed to operators that take a `while_waiting` argument
        self.while_waiting = None
        try:
            import inspect
            self.accepts_callback = 'while_waiting' in inspect.signature(solve).parameters
        except (TypeError, ValueError):
            self.accepts_callback = False

    def _children(self, parents, limit):
        """Footprints of the next ``depth`` generations of ``parents``, level by level; a level that would take the batch beyond
        ``limit`` candidates is dropped together with the deeper ones (all or nothing per level: a generation solved ahead in part
        still needs its own batch)."""
        found, level = {}, [frozenset(o.footprint) for o in parents]
        for _ in range(self.depth):
            nxt = []
            for _, fp, _ in _iterate_generation(level, self.adjacencies, self.max_seed_distance, ignored_cluster_labels=self.ignored, skip_last=True,
                                                memo=self.memo):
                if fp not in self.store:
                    nxt.append(fp)
                    if len(found) + len(nxt) > limit:
                        return found
            if not nxt:
                break
            for fp in nxt:
                found.setdefault(fp, None)
            level = nxt
        return found

    def __call__(self, objects, y, atoms, dsm_cfg, log_root_dir, status_line=DEFAULT_COMPUTING_STATUS_LINE, out=None, shard=None):
        objects = list(objects)
        keys = [frozenset(o.footprint) for o in objects]
        missing = [o for o, k in zip(objects, keys) if k not in self.store]
        if missing or self.batches == 0:
            asked = set(keys)
            ahead = []
            for fp in (self._children(objects, self.budget - len(missing)) if len(missing) < self.budget and self.depth > 0 else ()):
                if fp not in asked:
                    o = Object()
                    o.footprint = set(fp)
                    ahead.append(o)
            def renumber(error):                            # the index of a failed candidate counts within the batch the caller asked for
                if error.cidx is not None and error.cidx < len(missing):
                    error.cidx = next(i for i, o in enumerate(objects) if o is missing[error.cidx])
                return error
            extra = dict(while_waiting=self.while_waiting) if self.accepts_callback and self.while_waiting is not None else {}
            try:
                self.solve(missing + ahead, y, atoms, dsm_cfg, log_root_dir, status_line, out=out, shard=shard, **extra)
            except CvxprogError as error:
                if error.cidx is None or error.cidx < len(missing):
                    raise renumber(error)
                # a candidate solved ahead failed: the plain path might never have asked for it.  Solve what was asked for, alone.
                ahead = []
                self.depth = 0
                try:
                    self.solve(missing, y, atoms, dsm_cfg, log_root_dir, status_line, out=out, shard=shard)
                except CvxprogError as error2:
                    raise renumber(error2)
            self.batches += 1
            self.extra += len(ahead)
            for o in missing + ahead:
                self.store[frozenset(o.footprint)] = o
            self.unasked.update(frozenset(o.footprint) for o in ahead)
        for o, k in zip(objects, keys):
            src = self.store[k]
            if src is not o:
                # (the fragment array is shared, not copied as Object.set would: results are replaced, never modified in place)
                o.fg_offset, o._fg_fragment = src.fg_offset, src._fg_fragment      # (still lazy if nobody looked at it)
                o.energy, o.on_boundary, o.is_optimal, o.processing_time = src.energy, src.on_boundary, src.is_optimal, src.processing_time
                o.cvxprog_region_size = getattr(src, 'cvxprog_region_size', 0)
            if k in self.unasked:
                self.unasked.discard(k)
                self.served += 1


def _process_generation(cover, objects, previous_generation, y, atoms_map, adjacencies, dsm_cfg, max_seed_distance,
                        log_root_dir, pruning, ignored_cluster_labels, out, shard=None, solver=None, memo=None):
    """One generation: enumerate, prune by the energy bound, solve the survivors as ONE batch, keep those below
    their threshold (globalenergymin.py:326-368)."""
    new_objects, thresholds = [], []
    discarded = 0
    last_cluster, cluster_costs = None, None
    for parent, footprint, new_atom in _iterate_generation(previous_generation, adjacencies, max_seed_distance,
                                                           lambda c: c.footprint, ignored_cluster_labels, skip_last=True, memo=memo):
        cluster = adjacencies.get_cluster_label(next(iter(parent.footprint)))
        if cluster != last_cluster:
            last_cluster, cluster_costs = cluster, (cover.get_cluster_costs(cluster) if pruning == 'exact' else None)   # (isbi24 does not look at the cover)
        candidate = Object()
        candidate.footprint = footprint
        if pruning == 'exact':
            rest = adjacencies.get_atoms_in_cluster(cluster) - footprint
            rest_cost = sum(cover.get_atom(a).energy for a in rest)
            packing = sum(c.energy for c in solve_maxsetpack(
                [c for c in objects if c.is_optimal and c.footprint.issubset(footprint)], out=out.derive(muted=True)))
            lower = cover.beta + max(parent.energy + cover.get_atom(new_atom).energy, packing)
            upper = cluster_costs - rest_cost
            if upper < lower:
                discarded += 1
                continue
            thresholds.append(upper - cover.beta)
        elif pruning == 'isbi24':
            thresholds.append(parent.energy + cover.get_atom(new_atom).energy + cover.beta)
        else:
            raise ValueError(f'Unknown pruning mode "{pruning}"')
        new_objects.append(candidate)

    (solver or compute_objects)(new_objects, y, atoms_map, dsm_cfg, log_root_dir, out=out, shard=shard)

    next_generation = []
    for idx, (obj, threshold) in enumerate(zip(new_objects, thresholds)):
        if obj.energy < threshold:
            next_generation.append(obj)
        else:
            discarded += 1
            obj.fg_fragment = None                          # only footprint and energy are needed from now on
        obj.cidx = idx
    out.write(f'Next iteration: {len(next_generation)} ({discarded} discarded, {pruning} pruning)')
    return next_generation, new_objects


def _compute_generations(adjacencies, y_img, atoms_map, log_root_dir, pruning, dsm_cfg, beta=np.nan, max_iter=DEFAULT_MAX_ITER,
                         gamma=DEFAULT_GAMMA, max_seed_distance=np.inf, max_work_amount=DEFAULT_MAX_WORK_AMOUNT, out=None, shard=None, solver=None,
                         speculation=None, speculation_budget=DEFAULT_SPECULATION_BUDGET):
    """Returns ``(generations, costs, cover, objects, performance)`` (globalenergymin.py:183-271).  ``solver``: stands in for
    :func:`compute_objects` (same signature) -- the lock-step driver of :meth:`GlobalEnergyMinimization.process_many`.
    ``speculation``: generations solved ahead per batch (:class:`_Speculation`); ``None``: DEFAULT_SPECULATION (8) with the GPU operator, 0 when a test
    substitutes it or per-candidate log files are written (their names count within the reference's batches)."""
    out = get_output(out)
    solve = solver or compute_objects
    real_operator = solver is not None or getattr(compute_objects, '__module__', None) == Object.__module__
    if speculation is None:
        speculation = DEFAULT_SPECULATION if real_operator else 0
    ahead = None
    memo = {}                                               # expansions of footprints (skip_last=True), shared by every enumeration below
    if speculation > 0 and log_root_dir is None:
        ahead = _Speculation(solve, adjacencies, max_seed_distance, speculation, set(), speculation_budget, memo)
        solve = ahead

    atoms = []
    for label in adjacencies.atom_labels:
        obj = Object()
        obj.footprint = {label}
        atoms.append(obj)
    universes = []
    for cluster in adjacencies.cluster_labels:
        obj = Object()
        obj.footprint = set(adjacencies.get_atoms_in_cluster(cluster))
        universes.append(obj)
    out.write('\nIteration 1:')
    # the size of the search space (for the PerformanceReport) only needs the adjacency graph: counted while the first batch is on the GPU
    early = {}
    if ahead is not None and ahead.accepts_callback:
        def count_search_space():
            early['by_cluster'] = _remaining_by_cluster([atoms], adjacencies, max_seed_distance, max_amount=max_work_amount, skip_last=True, memo=memo)
        ahead.while_waiting = count_search_space
    if log_root_dir is None and (ahead is not None or (shard is None and getattr(compute_objects, '__module__', None) == Object.__module__)):   # (not when a test substitutes the operator)
        # the atoms and the cluster universes do not depend on each other (globalenergymin.py:192,199 computes them one after the
        # other): ONE batch of the engine -- a batch costs a round trip to the GPU whatever its size.  (With per-candidate log
        # files the two keep their own batches: their logs go to different directories.)
        try:
            solve(atoms + universes, y_img, atoms_map, dsm_cfg, None, ('Computing objects and universe costs', 'Computed objects and universe costs'), out=out, shard=shard)
        except CvxprogError as error:                       # the index of a failed candidate counts within its own batch (objects.py:309-318)
            if error.cidx is not None and error.cidx >= len(atoms):
                error.cidx -= len(atoms)
            raise
    else:
        solve(atoms, y_img, atoms_map, dsm_cfg, _generation_log_dir(log_root_dir, 1), out=out, shard=shard)
        solve(universes, y_img, atoms_map, dsm_cfg, _generation_log_dir(log_root_dir, 0),
              ('Computing universe costs', 'Universe costs computed'), out=out, shard=shard)
    if ahead is not None:
        ahead.while_waiting = None
    atom_by_label = {next(iter(a.footprint)): a for a in atoms}

    solved_directly, trivial = set(), set()        # Criterion 2 / universes of one or two atoms
    for cluster, universe in zip(adjacencies.cluster_labels, universes):
        if len(universe.footprint) <= 2:
            trivial.add(cluster)
        members = [atom_by_label[a] for a in adjacencies.get_atoms_in_cluster(cluster)]
        if all(m.is_optimal for m in members) and universe.energy <= beta + sum(m.energy for m in members):
            solved_directly.add(cluster)

    if ahead is not None:
        ahead.ignored.update(solved_directly)               # nothing is solved ahead in clusters that need no iterations
    cover = MinSetCover(atoms, beta, adjacencies, max_iter=max_iter, gamma=gamma)
    cover.update(universes, out.derive(muted=True))
    costs = [cover.costs]
    out.write(f'Solution costs: {costs[-1]:,g}')
    out.write(f'Clusters solved directly: {len(solved_directly)} / {len(adjacencies.cluster_labels)}')
    performance = PerformanceReport(direct_solution_trial_count=len(adjacencies.cluster_labels),
                                    direct_solution_success_count=len(solved_directly))

    generations = [atoms]
    objects = atoms + universes
    progress = lambda ignored: _estimate_progress(generations, adjacencies, max_seed_distance, max_amount=max_work_amount,
                                                  ignored_cluster_labels=ignored, skip_last=True, memo=memo)
    by_cluster = early['by_cluster'] if 'by_cluster' in early else _remaining_by_cluster(generations, adjacencies, max_seed_distance, max_amount=max_work_amount, skip_last=True, memo=memo)
    first = lambda ignored: _estimate_progress(generations, adjacencies, max_seed_distance, max_amount=max_work_amount,
                                               ignored_cluster_labels=ignored, skip_last=True, by_cluster=by_cluster)
    performance.nontrivial_object_count = first(trivial)[1]      # (one enumeration for both counts: a footprint stays in its cluster)
    performance.overall_object_count = performance.nontrivial_object_count + len(objects)
    performance.iterative_object_count = first(solved_directly)[1]
    performance.overall_computed_object_count = len(objects)

    # With 'isbi24' pruning the thresholds of a generation do not depend on the cover (globalenergymin.py:344-347), only the final
    # result does: its updates -- one min-weight set cover per touched cluster and generation -- are then applied, in order, while the
    # next batch is on the GPU instead of between the batches.  (Not with a log that shows the costs after every iteration, not with
    # 'exact' pruning, whose bounds read the cover, and not when the operator cannot call back.)
    pending = []

    def flush_cover():
        for generation in pending:
            cover.update(generation, out.derive(muted=True))
            costs.append(cover.costs)
        del pending[:]

    defer_cover = pruning == 'isbi24' and getattr(out, 'muted', False) and ahead is not None and ahead.accepts_callback
    if defer_cover:
        ahead.while_waiting = flush_cover
    if len(solved_directly) < len(adjacencies.cluster_labels):
        while True:
            number = 1 + len(generations)
            out.write('')
            if not getattr(out, 'muted', False):               # the estimate enumerates everything that is left: only for the log line
                done, todo = progress(solved_directly)         # (it cannot raise here: what is left only shrinks after the calls above)
                text = 'progress unknown' if np.isnan(done) or np.isnan(todo) else f'(finished {100 * done / (todo + done):.0f}% or more)'
                out.write(f'Iteration {number}: {Text.style(text, Text.BOLD)}')
            new_generation, new_objects = _process_generation(
                cover, objects, generations[-1], y_img, atoms_map, adjacencies, dsm_cfg, max_seed_distance,
                _generation_log_dir(log_root_dir, number), pruning, solved_directly, out, shard=shard, solver=solve, memo=memo)
            objects += new_objects
            performance.iterative_computed_object_count += len(new_objects)
            if not new_generation:
                break
            generations.append(new_generation)
            if defer_cover:                                 # applied, in order, while the next batch is on the GPU (or at the end)
                pending.append(new_generation)
            else:
                cover.update(new_generation, out.derive(muted=True))
                costs.append(cover.costs)
                out.write(f'Solution costs: {costs[-1]:,g}')
        flush_cover()

    performance.nontrivial_computed_object_count += performance.iterative_computed_object_count
    performance.overall_computed_object_count += performance.iterative_computed_object_count
    performance._assert_integrity()
    if ahead is not None:                                   # (not part of the reference's report)
        performance.engine_batches, performance.speculative_object_count = ahead.batches, ahead.extra - ahead.served   # solved ahead and never asked for
        out.write(f'Solved ahead: {ahead.extra} candidates in {ahead.batches} batches ({ahead.served} of them asked for later)')
    out.write('')
    out.write(f'Non-trivial pruning: {100 * performance.nontrivial_pruning_success:.1f}% '
              f'(computed {performance.nontrivial_computed_object_count} / {performance.nontrivial_object_count})')
    return generations, costs, cover, objects, performance



class _LockStep:
    """Rendezvous of the per-image threads of :meth:`GlobalEnergyMinimization.process_many`: ``submit`` has the signature of
    :func:`compute_objects`; when every thread that is still running has submitted a batch, all of them are solved together."""

    def __init__(self, n, out, stream=None):
        self.active, self.out, self.stream = n, out, stream   # stream: the torch stream this group's batches run on (None: the current one)
        self.jobs, self.round, self.error = [], 0, None      # error: a failure of the batch as a whole (every image); job_errors: of single images
        self.job_errors = {}
        self.cv = threading.Condition()
        self.batches = 0                      # multi-image batches solved (diagnostics / tests)

    def submit(self, objects, y, atoms, dsm_cfg, log_root_dir, status_line=None, out=None, shard=None, while_waiting=None):
        """``while_waiting``: host work of the calling thread that does not need the results (compute_objects): the thread that
        completes the rendezvous runs it after the batch has been launched, the others right after submitting -- so the batch
        starts as soon as the last image has its candidates, and whatever host work is left overlaps with it."""
        assert shard is None, 'process_many and sharded batches are separate ways to fill the GPUs'
        me = threading.get_ident()
        with self.cv:
            my_round = self.round
            self.jobs.append((list(objects), y, atoms, dsm_cfg, log_root_dir, me))
            last = len(self.jobs) >= self.active
            if last:
                self._flush(while_waiting)
        if not last and while_waiting is not None:
            while_waiting()                                  # (outside the lock)
        with self.cv:
            while self.round == my_round and self.error is None:
                self.cv.wait()
            if self.error is not None:
                raise self.error
            mine = self.job_errors.pop(me, None)             # a failed candidate of THIS image (CvxprogError, ...): the others go on
            if mine is not None:
                raise mine

    def leave(self):
        with self.cv:
            self.active -= 1
            if self.jobs and len(self.jobs) >= self.active:
                self._flush()

    def _flush(self, while_waiting=None):
        jobs, self.jobs = self.jobs, []
        try:
            cfg = jobs[0][3]
            assert all(j[3] == cfg for j in jobs), 'the images of one lock-step run share the dsm/* hyper-parameters'
            errors = [None] * len(jobs)
            if self.stream is None:
                compute_objects_multi([(j[0], j[1], j[2]) for j in jobs], cfg, [j[4] for j in jobs], out=self.out, while_waiting=while_waiting, errors=errors)
            else:
                import torch
                with torch.cuda.stream(self.stream):
                    compute_objects_multi([(j[0], j[1], j[2]) for j in jobs], cfg, [j[4] for j in jobs], out=self.out, while_waiting=while_waiting, errors=errors)
            for j, e in zip(jobs, errors):
                if e is not None:
                    self.job_errors[j[5]] = e
            self.batches += 1
        except BaseException as e:               # noqa: BLE001 -- handed to every waiting thread
            self.error = e
        self.round += 1
        self.cv.notify_all()


class GlobalEnergyMinimization(Stage):
    """Stage ``global-energy-minimization``.  ``shard`` (optional): a :class:`superdsm_amd.dist.Sharder` that splits
    every batch of candidates over the ranks of a process group."""

    ENABLED_BY_DEFAULT = True

    def __init__(self, shard=None):
        super().__init__('global-energy-minimization',
                         inputs=['y', 'y_mask', 'atoms', 'adjacencies', 'dsm_cfg'],
                         outputs=['y_img', 'cover', 'objects', 'performance'])
        self.shard = shard

    @staticmethod
    def _hyperparameters(cfg):
        pruning = cfg.get('pruning', 'exact')
        beta = cfg.get('beta', 0)
        max_iter = cfg.get('max_iter', DEFAULT_MAX_ITER)
        gamma = cfg.get('gamma', DEFAULT_GAMMA)
        max_seed_distance = cfg.get('max_seed_distance', np.inf)
        max_work_amount = cfg.get('max_work_amount', DEFAULT_MAX_WORK_AMOUNT)
        assert 0 < gamma < 1
        assert pruning in ('exact', 'isbi24')
        return pruning, beta, max_iter, gamma, max_seed_distance, max_work_amount, cfg.get('speculation', None)

    def process(self, input_data, cfg, out, log_root_dir, solver=None, speculation_budget=DEFAULT_SPECULATION_BUDGET):
        y_img = Image.create_from_array(input_data['y'], normalize=False, mask=input_data['y_mask'])
        y_img._sdsm_pinned = None                            # (set after the first batch: the device copy of THIS Image object serves all batches of the call)
        pruning, beta, max_iter, gamma, max_seed_distance, max_work_amount, speculation = self._hyperparameters(cfg)
        _, _, cover, objects, performance = _compute_generations(
            input_data['adjacencies'], y_img, input_data['atoms'], log_root_dir, pruning, dict(input_data['dsm_cfg']),
            beta, max_iter, gamma, max_seed_distance, max_work_amount, out, shard=self.shard, solver=solver, speculation=speculation,
            speculation_budget=speculation_budget)
        return {'y_img': y_img, 'cover': cover, 'objects': objects, 'performance': performance}

    def process_many(self, datas, cfg, out=None, log_root_dirs=None):
        """The stage for SEVERAL images in lock step (an image set as examples/NIH3T3, BASELINE.json configs[3]): generation k
        of every image is solved as ONE multi-image batch (objects.compute_objects_multi).  The generations of one image are
        sequential (globalenergymin.py:228-263) and hold only tens of candidates each -- far too few to fill a GPU; those of
        different images are independent.  Every image runs the unchanged host logic in a thread of its own; the threads
        meet whenever they need candidates solved.

        Used like ``Stage.__call__``, image by image: ``datas`` is a list of pipeline data dicts (the stage's inputs are read
        from, its outputs written to each), ``cfg`` one :class:`Config` for all or a list; returns the wall time.  The results
        equal those of calling the stage on every image separately."""
        datas = list(datas)
        cfgs = list(cfg) if isinstance(cfg, (list, tuple)) else [cfg] * len(datas)
        cfgs = [c.get(self.cfgns, {}) for c in cfgs]
        logs = list(log_root_dirs) if log_root_dirs is not None else [None] * len(datas)
        out = get_output(out)
        lock = _LockStep(len(datas), out)
        self.last_lockstep = lock
        locks, n_groups = [lock], 1
        # (Measured and dropped: two groups of images on streams of their own, the host logic of one overlapping the batch of the other
        # -- 13.5 vs 14.2 ms per image for 8 images: a batch of 4 images takes as long as one of 8, both as long as their largest region.)
        # generations are only solved ahead while the images together leave the GPU room for it; with many images in lock step the
        # batches fill it anyway and the host logic of the image threads is what takes the time (measured: 8 images, no gain)
        # generations are solved ahead while the images TOGETHER stay within what the GPU takes in one go: measured on 8 different
        # BBBC039-like images (round 3): none (12 dependent batches) 19.7 ms per image, 2048 candidates per batch 16.0, 4096 14.8 (4
        # batches), 8192 15.1
        budget = max(DEFAULT_SPECULATION_BUDGET, int(os.environ.get('SDSM_SPEC_BUDGET', DEFAULT_LOCKSTEP_BUDGET))) // len(datas)
        produced, errors = [None] * len(datas), [None] * len(datas)
        t0 = time.time()

        def work(i):
            try:
                stage_input = {inner: datas[i][outer] for outer, inner in self.inputs.items()}
                produced[i] = self.process(stage_input, cfgs[i], out.derive(muted=True), logs[i], solver=locks[i % n_groups].submit,
                                           speculation_budget=budget)
            except BaseException as e:            # noqa: BLE001 -- re-raised in the calling thread
                errors[i] = e
            finally:
                locks[i % n_groups].leave()

        # The image threads hand the interpreter lock to each other at every rendezvous; with CPython's default switch interval (5 ms)
        # a thread that wakes up may wait that long for the one that is computing -- as long as a whole batch takes on the GPU.
        # No cyclic garbage collection while the threads run: a full collection of a process that has PyTorch loaded takes tens of
        # milliseconds and stops all of them (measured: every other run of an 8-image set 6 ms per image slower).
        toggles = os.environ.get('SDSM_LOCKSTEP_TOGGLES', '1') != '0'      # (diagnostic: measure what the two settings below are worth)
        interval = sys.getswitchinterval()
        collecting = gc.isenabled()
        if toggles:
            sys.setswitchinterval(min(interval, 2e-4))
            gc.disable()
        try:
            threads = [threading.Thread(target=work, args=(i,), daemon=True) for i in range(len(datas))]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            if toggles:
                sys.setswitchinterval(interval)
                if collecting:
                    gc.enable()
        # an image whose candidate failed (CvxprogError) fails alone, as in the reference: the others' outputs are written, then the first
        # failure is raised (its ``image_index`` says which image)
        for data, prod, e in zip(datas, produced, errors):
            if e is not None:
                continue
            assert set(prod.keys()) == set(self.outputs), 'stage "%s" generated unexpected output' % self.name
            for inner, outer in self.outputs.items():
                data[outer] = prod[inner]
        failed = [i for i, e in enumerate(errors) if e is not None]
        if failed:
            e = errors[failed[0]]
            # a failure of the batch as a whole (not of one image's candidate) reaches every image thread as the SAME exception object:
            # it then carries all of their indices, not the first one
            shared = [i for i in failed if errors[i] is e]
            try:
                e.image_index = failed[0]
                e.image_indices = shared
            except AttributeError:                # (an exception type without a __dict__)
                pass
            raise e
        return time.time() - t0

    def configure_ex(self, scale, radius, diameter):
        return {
            'beta': (scale ** 2, 0.66),
            'max_seed_distance': (diameter, np.inf),
        }
