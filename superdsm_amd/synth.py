"""Deterministic synthetic workloads.

Test/bench infrastructure only: seeded nuclei images, a Voronoi-style partition into
atomic regions with 1-3 atoms per nucleus, and the candidate enumeration used by the
pure solver benchmark ("all connected atom subsets of size <= 3" plus the universes).
Nothing in here is on the product's compute path.
"""
import json
import math
import os

import numpy as np
import scipy.ndimage as ndi

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


def random_layout(shape, n, radius, seed, min_sep=0.6, border=None):
    """``n`` nuclei of nominal ``radius`` with centres >= ``min_sep * radius`` apart."""
    rng = np.random.default_rng(seed)
    border = radius if border is None else border
    centres = []
    tries = 0
    while len(centres) < n and tries < 200 * n:
        tries += 1
        c = np.array([rng.uniform(border, shape[0] - border), rng.uniform(border, shape[1] - border)])
        if all(np.hypot(*(c - o)) >= min_sep * radius for o in centres):
            centres.append(c)
    out = []
    for c in centres:
        f = rng.uniform(0.8, 1.25)
        out.append(dict(centre=(float(c[0]), float(c[1])), axes=(radius * f, radius / f),
                        angle=float(rng.uniform(0, math.pi)), amp=float(rng.uniform(0.6, 1.0))))
    return out


def bbbc039_like_layout(seed=1002, index=0):
    """Ellipses at the centres/areas of one reference BBBC039 regression CSV (data files: ``index`` 0 .. 7 picks one of eight of the
    198 per-image object tables, 68 .. 170 objects; 0 is the layout every test uses)."""
    if index == 0:
        with open(os.path.join(_DATA_DIR, 'bbbc039_like_layout.json')) as fp:
            spec = json.load(fp)
    else:
        with open(os.path.join(_DATA_DIR, 'bbbc039_like_layouts.json')) as fp:
            spec = json.load(fp)['layouts'][index]
    rng = np.random.default_rng(seed + 7919 * index)
    out = []
    for area, cx, cy in spec['objects']:
        r = math.sqrt(area / math.pi)
        f = rng.uniform(0.8, 1.25)
        out.append(dict(centre=(float(cy), float(cx)), axes=(r * f, r / f),
                        angle=float(rng.uniform(0, math.pi)), amp=float(rng.uniform(0.6, 1.0))))
    return tuple(spec['shape']), out


def render_image(shape, layout, seed, noise=0.02):
    """g = sum_k A_k exp(-1.2 d_k^2) + N(0, noise^2), min-max normalised to [0, 1]."""
    rng = np.random.default_rng(seed)
    g = np.zeros(shape)
    for nuc in layout:
        a, b = nuc['axes']
        ext = int(math.ceil(3.5 * max(a, b)))
        r0, c0 = nuc['centre']
        rs = slice(max(0, int(r0) - ext), min(shape[0], int(r0) + ext + 1))
        cs = slice(max(0, int(c0) - ext), min(shape[1], int(c0) + ext + 1))
        rr, cc = np.mgrid[rs, cs]
        ca, sa = math.cos(nuc['angle']), math.sin(nuc['angle'])
        dr, dc = rr - r0, cc - c0
        u = (ca * dr + sa * dc) / a
        v = (-sa * dr + ca * dc) / b
        g[rs, cs] += nuc['amp'] * np.exp(-1.2 * (u * u + v * v))
    g += noise * rng.standard_normal(shape)
    g -= g.min()
    g /= g.max()
    return g


def make_atoms(y, layout, seed):
    """Partition the image into atoms (1-3 per nucleus, seeded straight cuts = Voronoi of offset seeds).

    Returns ``(atoms int32 HxW with labels 1..A, clusters int32 HxW, seeds list[(r, c)])``.
    Atoms of one connected foreground component share a cluster label.
    """
    rng = np.random.default_rng(seed)
    fg = y > 0
    cc_labels, _ = ndi.label(fg)
    seeds = []
    for nuc in layout:
        k = int(rng.integers(1, 4))
        phi = rng.uniform(0, 2 * math.pi)
        rad = 0.45 * min(nuc['axes'])
        pts = [nuc['centre']] if k == 1 else [
            (nuc['centre'][0] + rad * math.cos(phi + 2 * math.pi * i / k),
             nuc['centre'][1] + rad * math.sin(phi + 2 * math.pi * i / k)) for i in range(k)]
        for p in pts:
            r, c = int(round(p[0])), int(round(p[1]))
            if 0 <= r < y.shape[0] and 0 <= c < y.shape[1] and fg[r, c] and (r, c) not in seeds:
                seeds.append((r, c))
    assert len(seeds) > 0, 'no seed landed on foreground'
    marker = np.ones(y.shape, bool)
    for r, c in seeds:
        marker[r, c] = False
    _, (ir, ic) = ndi.distance_transform_edt(marker, return_indices=True)
    seed_label = np.zeros(y.shape, np.int32)
    for i, (r, c) in enumerate(seeds):
        seed_label[r, c] = i + 1
    atoms = seed_label[ir, ic].astype(np.int32)
    cluster_of_atom = np.zeros(len(seeds) + 1, np.int32)
    for i, (r, c) in enumerate(seeds):
        cluster_of_atom[i + 1] = cc_labels[r, c]
    # relabel clusters to 1..K
    uniq = np.unique(cluster_of_atom[1:])
    remap = {int(u): i + 1 for i, u in enumerate(uniq)}
    cluster_of_atom[1:] = [remap[int(u)] for u in cluster_of_atom[1:]]
    clusters = cluster_of_atom[atoms]
    return atoms, clusters.astype(np.int32), seeds


def enumerate_candidates(adjacencies, max_size=3):
    """All connected atom subsets of size <= ``max_size`` plus every cluster's universe (deduplicated)."""
    seen = set()
    out = []

    def push(fp):
        fp = frozenset(fp)
        if fp not in seen:
            seen.add(fp)
            out.append(fp)

    frontier = [frozenset([a]) for a in sorted(adjacencies.atom_labels)]
    for fp in frontier:
        push(fp)
    for _ in range(max_size - 1):
        nxt = []
        for fp in frontier:
            nb = set()
            for a in fp:
                nb |= adjacencies[a]
            for a in sorted(nb - fp):
                g = fp | {a}
                if g not in seen:
                    push(g)
                    nxt.append(g)
        frontier = nxt
    for cl in sorted(adjacencies.cluster_labels):
        push(adjacencies.get_atoms_in_cluster(cl))
    return [sorted(fp) for fp in out]


# hyper-parameters per BASELINE.json config (derived-parameter table)
WORKLOADS = {
    'synthetic256': dict(shape=(256, 256), n=10, radius=15, seed=1001, scale=10),
    'synthetic512': dict(shape=(512, 512), n=60, radius=15, seed=1006, scale=10),     # BASELINE.json north_star: 'synthetic 512x512 nuclei images'
    'bbbc039_like': dict(seed=1002, scale=10),
    'gowt1_like':   dict(shape=(1024, 1024), n=25, radius=31, seed=1003, scale=42.43),
    'nih3t3_like':  dict(shape=(1344, 1024), n=48, radius=43, seed=1004, scale=40),
    'synthetic4096': dict(shape=(4096, 4096), n=2000, radius=15, seed=1005, scale=10),
}


def offset_image(g, sigma2, sigma1=math.sqrt(2), offset_clip=3):
    """Offset intensities y of a synthetic image via SciPy (test-data generation; same formula as the
    reference's preprocessing, superdsm/preprocess.py:42-64, so that scenes can be built without a GPU)."""
    off = ndi.gaussian_filter(g, sigma2)
    clip_abs = offset_clip * g.std()
    offc = ndi.gaussian_filter(g.clip(0, clip_abs), sigma2)
    t = ndi.distance_transform_edt(~(g > clip_abs))
    t = (sigma2 - t).clip(0, np.inf)
    t = (t / t.max()) ** 2
    return ndi.gaussian_filter(g, sigma1) - ((1 - t) * offc + t * off)


def dsm_config_for_scale(scale, alpha_factor=0.0005):
    """AF_ rule of the reference (superdsm/automation.py:71-77 with the factors of dsmcfg.py:91-97)."""
    return dict(scale=1000, epsilon=1.0, alpha=scale ** 2 * alpha_factor, smooth_amount=max(4, int(scale * 0.2)),
                smooth_subsample=max(8, int(scale * 0.4)), gaussian_shape_multiplier=2, background_margin=max(8, int(scale * 0.4)),
                init='elliptical')
