"""Adjacency graph of atomic image regions (host-side bookkeeping of the hot path's caller).

Same public behaviour as the reference's ``superdsm.atoms.AtomAdjacencyGraph`` (superdsm/atoms.py:19-291),
restated with vectorised 4-neighbour pixel pairs instead of per-atom morphological dilations (the reference
dilates every atom with ``disk(1)``, i.e. the 4-neighbourhood, atoms.py:53,62).
"""
import numpy as np


class AtomAdjacencyGraph:
    """``adj[a]`` is the set of atoms adjacent to atom ``a`` inside the foreground of ``a``'s cluster."""

    def __init__(self, atoms, clusters, fg_mask, seeds, out=None):
        atoms = np.asarray(atoms)
        clusters = np.asarray(clusters)
        fg_mask = np.asarray(fg_mask, bool)
        n = int(atoms.max()) if atoms.size else 0
        present = np.zeros(n + 1, bool)
        present[np.unique(atoms[atoms > 0])] = True
        # cluster of an atom = cluster label of its first pixel in raster order (atoms.py:59)
        flat_atoms, flat_clusters = atoms.reshape(-1), clusters.reshape(-1)
        first = np.full(n + 1, -1, np.int64)
        idx = np.nonzero(flat_atoms > 0)[0]
        order = idx[::-1]
        first[flat_atoms[order]] = order                # the last write wins -> smallest raster index
        self._cluster_by_atom = {int(a): int(flat_clusters[first[a]]) for a in range(1, n + 1) if present[a]}
        # Built with in-place unions in ascending label order, as the reference does (atoms.py:63-71): CPython lays a set
        # out differently for `|=` and `.add`, and the iteration order of these sets decides the floating-point
        # summation order of exact ties in the pruning bound (globalenergymin.py:341-346).
        self._atoms_by_cluster = {}
        for a, cl in self._cluster_by_atom.items():
            if cl not in self._atoms_by_cluster:
                self._atoms_by_cluster[cl] = set()
            self._atoms_by_cluster[cl] |= {a}
        self._adjacencies = {a: set() for a in range(1, n + 1)}
        cl_of = np.zeros(n + 1, np.int64)
        for a, cl in self._cluster_by_atom.items():
            cl_of[a] = cl
        for p, q in ((np.s_[:, :-1], np.s_[:, 1:]), (np.s_[:-1, :], np.s_[1:, :])):
            for src, dst in ((p, q), (q, p)):
                a, b = atoms[src], atoms[dst]
                ok = (a > 0) & (b > 0) & (a != b) & fg_mask[dst] & (clusters[dst] == cl_of[a])
                for x, y in set(zip(a[ok].tolist(), b[ok].tolist())):
                    self._adjacencies[x].add(y)
                    self._adjacencies[y].add(x)          # kept symmetric (atoms.py:68-69)
        self._seeds = {}
        seeds = [tuple(int(v) for v in s) for s in seeds]
        for a in self._cluster_by_atom:
            inside = [s for s in seeds if atoms[s] == a]
            assert len(inside) == 1, f'There is no (unique) seed. Number of possible seeds: {len(inside)}'
            self._seeds[a] = inside[0]

    def __getitem__(self, atom_label):
        return self._adjacencies[atom_label]

    def get_cluster_label(self, atom_label):
        return self._cluster_by_atom[atom_label]

    def get_atoms_in_cluster(self, cluster_label):
        return self._atoms_by_cluster[cluster_label]

    @property
    def cluster_labels(self):
        return frozenset(self._atoms_by_cluster.keys())

    @property
    def atom_labels(self):
        return frozenset(self._cluster_by_atom.keys())

    def get_seed(self, atom_label):
        return self._seeds[atom_label]

    def get_atom_degree(self, atom_label):
        return len(self[atom_label])

    @property
    def max_degree(self):
        return max(self.get_atom_degree(a) for a in self.atom_labels)

    def get_edge_lines(self, accept='all', reduce=True):
        if isinstance(accept, str) and accept == 'all':
            accept = lambda atom_label: True
        assert callable(accept), f'Not a callable: {accept}'
        lines = []
        for a in self.atom_labels:
            if not accept(a):
                continue
            for b in self[a]:
                if accept(b) and not (reduce and a > b):
                    lines.append((self.get_seed(a), self.get_seed(b)))
        return lines
