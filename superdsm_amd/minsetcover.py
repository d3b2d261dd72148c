"""Approximate min-weight set cover on the host (reference: superdsm/minsetcover.py:4-164, Algorithm 2 of
Kostrykin & Rohr, TPAMI 2023), restated.  Stays on the CPU by design (BASELINE.json north_star)."""
import ctypes

import numpy as np

from .output import get_output

DEFAULT_MAX_ITER = 5
DEFAULT_GAMMA = 0.8


def _greedy_cover(objects, beta):
    """Greedy phase: repeatedly accept the object with the lowest price (energy + beta) / #newly covered atoms;
    ties are resolved in favour of the object that comes first in ``objects``."""
    uncovered = set().union(*(c.footprint for c in objects))
    candidates = list(objects)
    accepted = []
    while candidates:
        best = min(candidates, key=lambda c: (c.energy + beta) / len(c.footprint & uncovered))
        accepted.append(best)
        uncovered -= best.footprint
        candidates = [c for c in candidates if c.footprint & uncovered]
    return accepted


def _merge_phase(objects, accepted, beta):
    """Merge phase: a not-yet-accepted object replaces the accepted objects it fully contains if that is cheaper.
    (Same decisions as minsetcover.py:4-21; membership by identity sets and subset / disjointness tests instead of building the
    intersections -- this loop is the host-side hot spot of the stage once the solves run on the GPU.)"""
    replaced = 0
    taken = {id(c) for c in accepted}
    for new in sorted((c for c in objects if id(c) not in taken), key=lambda c: c.energy + beta):
        nf = new.footprint
        inside = []
        for c in accepted:
            cf = c.footprint
            if cf.isdisjoint(nf):
                continue
            if not cf <= nf:
                inside = None                       # partial overlap: not a valid replacement
                break
            inside.append(c)
        if inside is None:
            continue
        if new.energy + beta < sum(c.energy + beta for c in inside):
            replaced += len(inside)
            gone = {id(c) for c in inside}
            accepted = [c for c in accepted if id(c) not in gone] + [new]
    return accepted, replaced


def _solve_once(objects, beta, merge, out):
    accepted = _greedy_cover(objects, beta)
    out.write(f'MINSETCOVER - GREEDY accepted objects: {len(accepted)}')
    if merge:
        accepted, replaced = _merge_phase(objects, accepted, beta)
        out.write(f'MINSETCOVER - MERGED objects: {replaced}')
    return accepted


def _bitsets(objects):
    """Footprints as bit sets over the atoms that occur: (uint64 array [n, words], words)."""
    index = {}
    ints = []
    for c in objects:
        m = 0
        for a in c.footprint:
            i = index.get(a)
            if i is None:
                i = index[a] = len(index)
            m |= 1 << i
        ints.append(m)
    words = max(1, (len(index) + 63) // 64)
    if words == 1:
        return np.array(ints, np.uint64).reshape(len(objects), 1), 1
    masks = np.zeros((len(objects), words), np.uint64)
    for i, m in enumerate(ints):
        for w in range(words):
            masks[i, w] = (m >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return masks, words


def solve_minsetcover(objects, beta, merge=True, max_iter=DEFAULT_MAX_ITER, gamma=DEFAULT_GAMMA, out=None):
    """Cover for ``beta``; retried with ``beta * gamma`` (up to ``max_iter`` levels), keeping a retry only if it is
    cheaper when priced with the ORIGINAL beta of the level that spawned it (minsetcover.py:53-88).  Runs as native host code
    (sdsm_minsetcover, same decisions; :func:`solve_minsetcover_py` is the line-by-line restatement it is tested against)."""
    assert beta >= 0 and 0 < gamma < 1
    objects = list(objects)
    if not objects:
        return []
    from . import _capi
    L = _capi.lib()
    masks, words = _bitsets(objects)
    energies = np.array([c.energy for c in objects], np.float64)
    if not np.isfinite(energies).all():
        return solve_minsetcover_py(objects, beta, merge, max_iter, gamma, out)
    sel = np.zeros(len(objects) + 1, np.int32)              # (the count of selected objects in the last element)
    ptr = lambda a: a.__array_interface__['data'][0]         # (ctypes' data_as costs microseconds per call; this runs a hundred times per image)
    code = L.sdsm_minsetcover(len(objects), words, ptr(masks), ptr(energies), float(beta), int(bool(merge)), int(max_iter), float(gamma), ptr(sel),
                              ptr(sel) + 4 * len(objects))
    assert code == 0, 'sdsm_minsetcover: bad argument'
    nsel = int(sel[-1])
    if out is not None and out != 'muted' and not getattr(out, 'muted', False):
        get_output(out).write(f'MINSETCOVER accepted objects: {nsel}')
    return [objects[i] for i in sel[:nsel].tolist()]


def solve_minsetcover_py(objects, beta, merge=True, max_iter=DEFAULT_MAX_ITER, gamma=DEFAULT_GAMMA, out=None):
    """The same in Python, statement by statement after the reference."""
    assert beta >= 0 and 0 < gamma < 1
    out = get_output(out)
    solution = _solve_once(objects, beta, merge, out)
    if max_iter > 1 and beta > 0:
        out.write(f'MINSETCOVER retry with lower beta: {beta * gamma:g}')
        retry = solve_minsetcover_py(objects, beta * gamma, merge, max_iter - 1, gamma, out)
        price = lambda sol: sum(c.energy for c in sol) + beta * len(sol)
        if price(retry) < price(solution):
            return retry
    return solution


class MinSetCover:
    """Incrementally maintained per-cluster covers (minsetcover.py:91-164).  The footprints of a cluster's objects are kept as bit
    sets (appended to, never rebuilt) and the covers of ALL clusters an update touches are solved by one native call
    (sdsm_minsetcover_multi): the stage updates the cover once per generation, which touches a dozen clusters -- a hundred solves
    per image, each of which used to rebuild its bit sets and cross the foreign-function interface on its own."""

    def __init__(self, atoms, beta, adjacencies, **solve_minsetcover_kwargs):
        label_of = lambda atom: next(iter(atom.footprint))
        for atom in atoms:
            assert len(atom.footprint) == 1
        self.atoms = {label_of(atom): atom for atom in atoms}
        self.beta = beta
        self.adjacencies = adjacencies
        self.solve_minsetcover_kwargs = solve_minsetcover_kwargs
        self.objects_by_cluster = {cl: [] for cl in adjacencies.cluster_labels}
        self._bits = {cl: ({}, [], []) for cl in adjacencies.cluster_labels}     # cluster -> (atom -> bit, footprints as ints, energies)
        for a in atoms:                                     # (one pass; the reference filters the atoms once per cluster: same lists, same order)
            self._append(adjacencies.get_cluster_label(label_of(a)), a)
        self.solution_by_cluster = {cl: self.objects_by_cluster[cl] for cl in adjacencies.cluster_labels}

    def _append(self, cl, obj):
        self.objects_by_cluster[cl].append(obj)
        index, masks, energies = self._bits[cl]
        m = 0
        for a in obj.footprint:
            i = index.get(a)
            if i is None:
                i = index[a] = len(index)
            m |= 1 << i
        masks.append(m)
        energies.append(obj.energy)

    def get_atom(self, atom_label):
        return self.atoms[atom_label]

    def update(self, new_objects, out=None):
        touched = []
        for obj in new_objects:
            cl = self.adjacencies.get_cluster_label(next(iter(obj.footprint)))
            self._append(cl, obj)
            if cl not in touched:
                touched.append(cl)
        if not touched:
            return
        kw = self.solve_minsetcover_kwargs
        merge, max_iter, gamma = kw.get('merge', True), kw.get('max_iter', DEFAULT_MAX_ITER), kw.get('gamma', DEFAULT_GAMMA)
        native = []
        for cl in touched:                                   # objects without energies yet / non-finite ones: the Python restatement
            objs = self.objects_by_cluster[cl]
            index, masks, _ = self._bits[cl]
            # The reference reads `energy` and `footprint` of every object when it solves (minsetcover.py:24-36), not when the object was
            # added: energies are taken afresh from the objects here, and objects that reached the public list by another way than
            # update() get their bit sets now
            for obj in objs[len(masks):]:
                m = 0
                for a in obj.footprint:
                    m |= 1 << index.setdefault(a, len(index))
                masks.append(m)
            assert len(masks) == len(objs), 'objects were removed from objects_by_cluster'
            self._bits[cl] = (index, masks, [o.energy for o in objs])
            e = self._bits[cl][2]
            if all(v is not None and v == v and abs(v) != float('inf') for v in e):
                native.append(cl)
            else:
                self.solution_by_cluster[cl] = solve_minsetcover(self.objects_by_cluster[cl], self.beta, out=out, **kw)
        if not native:
            return
        from . import _capi
        L = _capi.lib()
        ns = np.array([len(self._bits[cl][1]) for cl in native], np.int32)
        words = np.array([max(1, (len(self._bits[cl][0]) + 63) // 64) for cl in native], np.int32)
        if int(words.max()) == 1:
            masks = np.array([m for cl in native for m in self._bits[cl][1]], np.uint64)
        else:
            parts = []
            for cl, w in zip(native, words.tolist()):
                for m in self._bits[cl][1]:
                    parts.extend((m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(w))
            masks = np.array(parts, np.uint64)
        energies = np.array([v for cl in native for v in self._bits[cl][2]], np.float64)
        sel = np.zeros(int(ns.sum()), np.int32)
        nsel = np.zeros(len(native), np.int32)
        ptr = lambda a: a.__array_interface__['data'][0]
        code = L.sdsm_minsetcover_multi(len(native), ptr(ns), ptr(words), ptr(masks), ptr(energies), float(self.beta), int(bool(merge)), int(max_iter), float(gamma),
                                        ptr(sel), ptr(nsel))
        assert code == 0, 'sdsm_minsetcover_multi: bad argument'
        pos = 0
        sel_l, nsel_l = sel.tolist(), nsel.tolist()
        for cl, n, k in zip(native, ns.tolist(), nsel_l):
            objs = self.objects_by_cluster[cl]
            self.solution_by_cluster[cl] = [objs[i] for i in sel_l[pos:pos + k]]
            pos += n

    def get_cluster_costs(self, cluster_label):
        sol = self.solution_by_cluster[cluster_label]
        return sum(c.energy for c in sol) + self.beta * len(sol)

    @property
    def solution(self):
        return [c for sol in self.solution_by_cluster.values() for c in sol]

    @property
    def costs(self):
        sol = self.solution
        return sum(c.energy for c in sol) + self.beta * len(sol)
