"""Host-side mirror of the reference's plugin API against literal known-answer tests of the reference
(tests/test_objects.py, tests/test_image.py, tests/test_atoms.py) and golden vectors produced by running the
reference (tests/golden/config.json, setcover.json).  CPU only; the GPU solver is replaced by the fixture's energy table."""
import json
import os

import numpy as np
import pytest

from superdsm_amd import atoms as sd_atoms
from superdsm_amd import automation, config, globalenergymin, image, maxsetpack, minsetcover, objects, pipeline
from superdsm_amd.dsmcfg import DSM_Config

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


# ---- reference KATs: tests/test_objects.py:11-50 ---------------------------------------------------------
def test_fill_foreground_kat():
    obj = objects.BaseObject()
    obj.fg_fragment = np.array([[False, True], [True, True], [True, False]])
    obj.fg_offset = (1, 2)
    actual = np.zeros((4, 5), bool)
    obj.fill_foreground(actual)
    expected = np.array([[0, 0, 0, 0, 0], [0, 0, 0, 1, 0], [0, 0, 1, 1, 0], [0, 0, 1, 0, 0]], bool)
    np.testing.assert_array_equal(actual, expected)


def test_get_mask_kat():
    atoms = np.array([[1, 1, 2], [1, 3, 2], [3, 3, 3]])
    obj = objects.Object()
    obj.footprint = {2, 3}
    np.testing.assert_array_equal(obj.get_mask(atoms), np.array([[0, 0, 1], [0, 1, 1], [1, 1, 1]], bool))


def test_extract_foreground_fragment_kat():
    mask = np.array([[0, 0, 0, 0, 0], [0, 0, 0, 1, 0], [0, 0, 1, 1, 0], [0, 0, 1, 0, 0]], bool)
    off, frag = objects.extract_foreground_fragment(mask)
    np.testing.assert_array_equal(off, [1, 2])
    np.testing.assert_array_equal(frag, np.array([[0, 1], [1, 1], [1, 0]], bool))
    off, frag = objects.extract_foreground_fragment(np.zeros((3, 3), bool))
    np.testing.assert_array_equal(off, [0, 0])
    np.testing.assert_array_equal(frag, [[False]])


def test_fragments_unpacked_on_demand_equal_the_batch_unpack():
    """compute_objects keeps the bit-packed masks of a batch on the host (engine.PackedFragments) and Object.fg_fragment unpacks one
    fragment when it is looked at: the same arrays as unpacking the whole batch (objects.py:148-174 applied to the region-bbox masks),
    and independent of the buffers the batch was read from (staging buffers are reused by the next batch)."""
    from superdsm_amd import _capi, engine
    rng = np.random.default_rng(5)
    n = 7
    recs = np.zeros(n, _capi.RECORD_DTYPE)
    info = np.zeros((n, 4), np.int32)
    offs = np.zeros(n, np.int64)
    words, expect = [], []
    for i in range(n):
        h, w = int(rng.integers(3, 9)), int(rng.integers(3, 40))
        r0, c0 = int(rng.integers(0, 50)), int(rng.integers(0, 50))
        box = rng.random((h, w)) < 0.5
        box[0, :] = box[-1, :] = False                           # the fragment is the bounding box of the mask inside its box
        box[1, 1] = box[h - 2, w - 2] = True
        rr, cc = np.nonzero(box)
        frag = box[rr.min():rr.max() + 1, cc.min():cc.max() + 1]
        info[i] = (r0, c0, h, w)
        offs[i] = 4 * len(words)
        bits = np.zeros(((h * w + 31) // 32) * 32, np.uint8)
        bits[:h * w] = box.reshape(-1)
        words += list(np.packbits(bits.reshape(-1, 8), axis=1, bitorder='little').reshape(-1).view(np.uint32))
        recs[i]['fg_r0'], recs[i]['fg_c0'], recs[i]['fg_h'], recs[i]['fg_w'] = r0 + rr.min(), c0 + cc.min(), frag.shape[0], frag.shape[1]
        expect.append(((r0 + rr.min(), c0 + cc.min()), frag))
    recs[3]['status'] = _capi.CAND_TRIVIAL                         # no foreground: [[False]] at the origin (objects.py:172-174)
    expect[3] = ((0, 0), np.zeros((1, 1), bool))
    masks = np.array(words, np.uint32).view(np.uint8).copy()
    eager = engine.fragments_from_masks(recs, info, offs, masks)
    lazy = engine.fragments_from_masks(recs, info, offs, masks, lazy=True)
    masks[:] = 0xff                                                 # the staging buffer is reused by the next batch
    recs['fg_h'] = 1
    for i in range(n):
        obj = objects.Object()
        obj.fg_offset, obj.fg_fragment = lazy[i]
        assert type(obj._fg_fragment) is tuple                      # still packed
        for off, frag in (eager[i], (obj.fg_offset, obj.fg_fragment)):
            assert tuple(int(v) for v in off) == expect[i][0]
            np.testing.assert_array_equal(frag, expect[i][1])
        assert isinstance(obj._fg_fragment, np.ndarray) and obj.fg_fragment is obj.fg_fragment


# ---- reference KATs: tests/test_image.py:10-40 -------------------------------------------------------------
def test_get_pixel_map_kat():
    expected = np.array([np.repeat(np.arange(5.0)[:, None], 5, 1), np.repeat(np.arange(5.0)[None, :], 5, 0)])
    np.testing.assert_allclose(image.get_pixel_map((5, 5)), expected)
    np.testing.assert_allclose(image.get_pixel_map((5, 5), normalized=True), expected / 4)


def test_bbox_kat():
    mask = np.array([[0, 0, 0, 0, 0], [0, 0, 0, 1, 0], [0, 0, 1, 1, 0], [0, 0, 1, 0, 0]], bool)
    b1, s1 = image.bbox(mask)
    b2, s2 = image.bbox(mask, include_end=True)
    np.testing.assert_array_equal(b1, [[1, 4], [2, 4]])
    np.testing.assert_array_equal(b2, [[1, 3], [2, 3]])
    assert s1 == (slice(1, 4, None), slice(2, 4, None)) and s2 == (slice(1, 3, None), slice(2, 3, None))


# ---- reference KATs: tests/test_atoms.py:12-69 -------------------------------------------------------------
@pytest.fixture(scope='module')
def kat_graph():
    atoms = np.array([[1, 1, 2, 4], [1, 3, 2, 4], [3, 3, 3, 4]])
    clusters = np.array([[1, 1, 2, 2], [1, 2, 2, 2], [2, 2, 2, 2]])
    fg = np.array([[1, 0, 1, 0], [1, 0, 1, 1], [1, 1, 1, 1]], bool)
    seeds = [(0, 0), (0, 2), (2, 1), (1, 3)]
    return sd_atoms.AtomAdjacencyGraph(atoms, clusters, fg, seeds), sd_atoms.AtomAdjacencyGraph(atoms, clusters, fg, seeds[::-1])


def test_adjacency_graph_kat(kat_graph):
    adj, adj_rev = kat_graph
    assert adj[1] == set() and adj[2] == {3, 4} and adj[3] == {2, 4} and adj[4] == {2, 3}
    assert adj.atom_labels == frozenset({1, 2, 3, 4}) and adj.cluster_labels == frozenset({1, 2})
    assert [adj.get_atom_degree(a) for a in (1, 2, 3, 4)] == [0, 2, 2, 2] and adj.max_degree == 2
    assert adj.get_atoms_in_cluster(1) == {1} and adj.get_atoms_in_cluster(2) == {2, 3, 4}
    assert [adj.get_cluster_label(a) for a in (1, 2, 3, 4)] == [1, 2, 2, 2]
    assert adj.get_edge_lines() == [((0, 2), (2, 1)), ((0, 2), (1, 3)), ((2, 1), (1, 3))]
    assert adj.get_edge_lines(lambda i: i != 4) == [((0, 2), (2, 1))]
    assert adj.get_edge_lines(lambda i: i != 4, reduce=False) == [((0, 2), (2, 1)), ((2, 1), (0, 2))]
    for g in (adj, adj_rev):
        assert [g.get_seed(a) for a in (1, 2, 3, 4)] == [(0, 0), (0, 2), (2, 1), (1, 3)]


# ---- Config semantics (superdsm/config.py) -----------------------------------------------------------------
def test_config_semantics():
    cfg = config.Config()
    assert cfg.get('a/b/c', 5) == 5 and cfg.entries == {'a': {'b': {'c': 5}}}          # get inserts the default
    cfg['a/b/d'] = None
    assert cfg.set_default('a/b/d', 7) is None and cfg.set_default('a/b/d', 7, override_none=True) == 7
    assert 'a/b' in cfg and 'a/x' not in cfg and cfg['a/b/c'] == 5
    other = cfg.derive(config.Config({'a': {'b': {'c': 6}}, 'z': 1}))
    assert other['a/b/c'] == 6 and other['z'] == 1 and cfg['a/b/c'] == 5
    assert cfg.pop('a/b/c', None) == 5 and 'a/b/c' not in cfg
    assert config.Config(cfg).entries is not cfg.entries and config.Config(cfg.entries).entries is cfg.entries
    assert cfg.md5.hexdigest() == config.Config(cfg).md5.hexdigest()


def test_af_expansion_matches_reference():
    """AF_ rule (automation.py:71-102) on the dataset task specs, against the reference's own create_config."""
    from superdsm_amd.globalenergymin import GlobalEnergyMinimization
    from superdsm_amd.preprocess import Preprocessing
    golden = json.load(open(os.path.join(G, 'config.json')))
    pl = pipeline.Pipeline()
    for st in (Preprocessing(), DSM_Config(), GlobalEnergyMinimization()):
        pl.append(st)
    for key, case in golden.items():
        cfg, scale = automation.create_config(pl, config.Config(json.loads(json.dumps(case['base']))))
        assert scale == case['scale']
        assert json.loads(json.dumps(cfg.entries)) == case['expanded'], key


def test_pipeline_ordering_and_contract():
    class A(pipeline.Stage):
        def __init__(self):
            super().__init__('a', inputs=['g_raw'], outputs=['x'])

        def process(self, input_data, cfg, out, log_root_dir):
            return {'x': input_data['g_raw'].sum()}

    class B(pipeline.Stage):
        def __init__(self):
            super().__init__('b', inputs=['x'], outputs=['z'])

        def process(self, input_data, cfg, out, log_root_dir):
            return {'z': input_data['x'] * cfg.get('factor', 2)}

    pl = pipeline.create_pipeline([B(), A()])
    assert [s.name for s in pl.stages] == ['a', 'b']
    data, cfg, timings = pl.process_image(np.array([[0.0, 2.0], [4.0, 4.0]]), config.Config({'b': {'factor': 3}}), out='muted')
    assert data['z'] == pytest.approx(3 * 2.5) and set(timings) == {'a', 'b'}      # g_raw is min-max normalised first
    data2, _, t2 = pl.process_image(None, config.Config(), first_stage='b', data=dict(data), out='muted')
    assert set(t2) == {'b'} and data2['z'] == pytest.approx(2 * 2.5)
    with pytest.raises(ValueError):
        pipeline.create_pipeline([B()])
    from superdsm_amd.globalenergymin import GlobalEnergyMinimization
    st = GlobalEnergyMinimization()
    assert st.name == 'global-energy-minimization' and set(st.inputs) == {'y', 'y_mask', 'atoms', 'adjacencies', 'dsm_cfg'}
    assert set(st.outputs) == {'y_img', 'cover', 'objects', 'performance'}
    assert st.configure(40) == {'beta': (1600, 0.66), 'max_seed_distance': (pytest.approx(2 * 40 * np.sqrt(2)), np.inf)}


# ---- set cover / set packing / generation logic against the reference run on a fake energy table ------------
@pytest.fixture(scope='module')
def setcover_fixture():
    d = json.load(open(os.path.join(G, 'setcover.json')))
    scene = np.load(os.path.join(G, 'setcover_scene.npz'))
    shape = tuple(int(v) for v in scene['shape'])
    fg = np.unpackbits(scene['fg'])[:shape[0] * shape[1]].reshape(shape).astype(bool)
    adj = sd_atoms.AtomAdjacencyGraph(scene['atoms'], scene['clusters'], fg, [tuple(s) for s in scene['seeds']])
    return d, adj, scene['atoms']


def test_adjacency_matches_reference(setcover_fixture):
    d, adj, _ = setcover_fixture
    ref = d['adjacency']
    assert sorted(adj.atom_labels) == ref['atom_labels'] and sorted(adj.cluster_labels) == ref['cluster_labels']
    for a in ref['atom_labels']:
        assert sorted(adj[a]) == ref['adjacency'][str(a)]
        assert adj.get_cluster_label(a) == ref['cluster_of_atom'][str(a)]
        assert list(adj.get_seed(a)) == ref['seeds'][str(a)]


class _Fake:
    def __init__(self, fp, energy):
        self.footprint, self.energy, self.is_optimal = set(fp), energy, True


def test_minsetcover_and_maxsetpack_match_reference(setcover_fixture):
    d, adj, _ = setcover_fixture
    from superdsm_amd import synth
    table = d['energy_table']
    cands = [fp for fp in synth.enumerate_candidates(adj, max_size=4) if len(fp) <= 3]
    fam = [_Fake(fp, table[','.join(map(str, fp))]) for fp in cands]
    for key, expected in d['minsetcover'].items():
        sol = minsetcover.solve_minsetcover(fam, float(key[4:]), out='muted')
        assert sorted(sorted(c.footprint) for c in sol) == expected
    assert sorted(sorted(c.footprint) for c in maxsetpack.solve_maxsetpack(fam, out='muted')) == d['maxsetpack']


@pytest.mark.parametrize('case', ['beta0_exact', 'beta0_isbi24', 'beta60_exact', 'beta60_isbi24', 'beta150_exact', 'beta150_isbi24'])
def test_generations_match_reference(setcover_fixture, case, monkeypatch):
    """Same batches, same pruning decisions, same cover and the same PerformanceReport as the reference's
    _compute_generations when both are fed the same energies."""
    d, adj, atoms = setcover_fixture
    table = d['energy_table']
    expected = d['generations'][case]
    beta, pruning = float(case.split('_')[0][4:]), case.split('_')[1]
    batches = []

    def fake_compute_objects(objs, y, atoms_map, dsm_cfg, log_root_dir, status_line=None, out=None, shard=None):
        objs = list(objs)
        batches.append([sorted(int(a) for a in o.footprint) for o in objs])
        for o in objs:
            o.energy = table[','.join(map(str, sorted(o.footprint)))]
            o.is_optimal, o.on_boundary, o.processing_time = True, False, 0
            o.fg_offset, o.fg_fragment = np.zeros(2, int), np.zeros((1, 1), bool)

    monkeypatch.setattr(globalenergymin, 'compute_objects', fake_compute_objects)
    gens, costs, cover, objs, perf = globalenergymin._compute_generations(adj, None, atoms, None, pruning, {}, beta=beta, out='muted')
    assert [sorted(b) for b in batches] == [sorted(b) for b in expected['batches']]
    np.testing.assert_allclose(costs, expected['costs'], rtol=1e-12)
    assert sorted(sorted(int(a) for a in o.footprint) for o in cover.solution) == expected['solution']
    assert len(objs) == expected['n_objects']
    assert {k: int(getattr(perf, k)) for k in perf.attributes} == expected['performance']
    assert [sorted(sorted(int(a) for a in o.footprint) for o in g) for g in gens] == [sorted(g) for g in expected['generations']]


def test_native_set_cover_and_packing_make_the_decisions_of_the_python_restatement():
    """sdsm_minsetcover / sdsm_maxsetpack (host C++) against the statement-by-statement Python restatements on random families:
    overlapping footprints over up to 150 atoms (several 64-bit words), ties in the energies, every beta / merge / level count."""
    from superdsm_amd import maxsetpack, minsetcover, objects
    rng = np.random.default_rng(12)
    for trial in range(60):
        n_atoms = int(rng.integers(3, 150))
        n = int(rng.integers(1, 60))
        fam = []
        for a in range(1, n_atoms + 1):                      # every atom is covered by its singleton
            o = objects.Object()
            o.footprint = frozenset([a])
            o.energy = float(np.round(rng.uniform(5, 60), 1))
            fam.append(o)
        for _ in range(n):
            o = objects.Object()
            start = int(rng.integers(1, n_atoms + 1))
            o.footprint = frozenset(range(start, min(n_atoms, start + int(rng.integers(1, 6))) + 1))
            o.energy = float(np.round(rng.uniform(5, 200), 0))          # rounded: ties do occur
            fam.append(o)
        rng.shuffle(fam)
        for beta in (0.0, 7.5, 80.0):
            for merge in (True, False):
                for max_iter in (1, 5):
                    a = minsetcover.solve_minsetcover(fam, beta, merge=merge, max_iter=max_iter, out='muted')
                    b = minsetcover.solve_minsetcover_py(fam, beta, merge=merge, max_iter=max_iter, out='muted')
                    assert [id(o) for o in a] == [id(o) for o in b], (trial, beta, merge, max_iter)
        a, b = maxsetpack.solve_maxsetpack(fam, out='muted'), maxsetpack.solve_maxsetpack_py(fam, out='muted')
        assert [id(o) for o in a] == [id(o) for o in b]


@pytest.mark.parametrize('pruning', ['isbi24', 'exact'])
def test_generations_solved_ahead_change_nothing_but_the_number_of_batches(pruning, monkeypatch):
    """globalenergymin._Speculation: every batch also solves the children of its candidates; the host logic must see the same
    candidates with the same results (generations, objects in order, costs, cover, PerformanceReport) in fewer batches, and a
    failure of a candidate that was only solved ahead must not surface."""
    import hashlib
    from superdsm_amd import objects, testing
    scene = testing.make_scene('bbbc039_like', max_size=3)
    adj = scene['adjacencies']
    h = lambda s: int(hashlib.sha1(s.encode()).hexdigest()[:8], 16) / 2 ** 32

    def energy(fp):
        fp = sorted(int(a) for a in fp)
        base = sum(30 + 20 * h(f'a{a}') for a in fp)
        return base if len(fp) == 1 else base * (0.75 + 0.5 * h(','.join(map(str, fp))))

    waited = []

    def run(depth, poison=None, budget=768, callback=False):
        calls = []

        def fake(objs, y, atoms_map, dsm_cfg, log_root_dir, status_line=None, out=None, shard=None, **kwargs):
            objs = list(objs)
            calls.append([frozenset(o.footprint) for o in objs])
            if kwargs.get('while_waiting') is not None:
                waited.append(len(calls))
                kwargs['while_waiting']()
            for k, o in enumerate(objs):
                if poison is not None and frozenset(o.footprint) == poison:
                    raise objects.CvxprogError('convex programming failed for the elliptical model', cidx=k)
                o.energy = energy(o.footprint)
                o.is_optimal, o.on_boundary, o.processing_time = True, False, 0
                o.fg_offset, o.fg_fragment = np.zeros(2, int), np.zeros((1, 1), bool)

        if callback:                                          # an operator that takes `while_waiting`, as objects.compute_objects does
            def fake_cb(objs, y, atoms_map, dsm_cfg, log_root_dir, status_line=None, out=None, shard=None, while_waiting=None):
                return fake(objs, y, atoms_map, dsm_cfg, log_root_dir, status_line, out=out, shard=shard, while_waiting=while_waiting)
        monkeypatch.setattr(globalenergymin, 'compute_objects', fake_cb if callback else fake)
        gens, costs, cover, objs, perf = globalenergymin._compute_generations(adj, None, scene['atoms'], None, pruning, {}, beta=5.0, out='muted', speculation=depth,
                                                                              speculation_budget=budget)
        state = ([sorted(sorted(o.footprint) for o in g) for g in gens], [sorted(o.footprint) for o in objs], [o.energy for o in objs], list(costs),
                 sorted(sorted(int(a) for a in o.footprint) for o in cover.solution), {k: int(getattr(perf, k)) for k in perf.attributes})
        return state, calls, perf

    plain, plain_calls, _ = run(0)
    assert len(plain[0]) >= 3, 'the fake energies must make the stage iterate'
    asked = set().union(*[set(c) for c in plain_calls])
    for depth in (1, 2):
        state, calls, perf = run(depth)
        assert state == plain
        assert len(calls) < len([c for c in plain_calls if c]) and perf.engine_batches == len(calls)
        assert perf.speculative_object_count == sum(len(c) for c in calls) - sum(len(c) for c in plain_calls)
    # the set-cover updates of a generation applied while the next batch is "on the GPU" (an operator with `while_waiting`; isbi24 only)
    state, calls, _ = run(1, callback=True)
    assert state == plain
    assert waited[0] == 1 and (len(waited) > 1) == (pruning == 'isbi24')      # the first batch: the search-space count; later ones: cover updates
    # a batch is only extended within the budget (what the GPU runs at once): all or nothing
    state, calls, _ = run(1, budget=60)
    assert state == plain and all(len(c) <= 60 or c in plain_calls or c == plain_calls[0] + plain_calls[1] for c in calls)
    assert len(calls) < len([c for c in plain_calls if c])
    # a candidate that the plain run never asks for fails: the run must not notice
    state, calls, _ = run(1)
    never = next(fp for c in calls for fp in c if fp not in asked)
    state, calls, _ = run(1, poison=never)
    assert state == plain
    # a candidate the plain run does ask for fails: the error surfaces with its index in the batch the caller asked for
    wanted = plain_calls[2][len(plain_calls[2]) // 2]
    with pytest.raises(objects.CvxprogError) as plain_error:
        run(0, poison=wanted)
    with pytest.raises(objects.CvxprogError) as ahead_error:
        run(1, poison=wanted)
    assert plain_calls[2][plain_error.value.cidx] == wanted
    assert ahead_error.value.cidx is not None


@pytest.mark.parametrize('workload', ['bbbc039_like', 'nih3t3_like', 'synthetic256'])
def test_native_search_space_count_equals_the_enumeration(workload):
    """sdsm_count_growth (host C++, bit sets per cluster) against the footprint-by-footprint enumeration of _iterate_generation
    that the reference's _estimate_progress does: same count per cluster, with and without the universe, for every seed distance."""
    import unittest.mock as mock
    from superdsm_amd import objects, testing
    scene = testing.make_scene(workload, max_size=2)
    adj = scene['adjacencies']
    atoms = []
    for label in adj.atom_labels:
        o = objects.Object()
        o.footprint = {label}
        atoms.append(o)
    assert globalenergymin._count_growth_native(adj, set(adj.atom_labels), np.inf, True, 10 ** 6) is not None
    for max_seed_distance in (np.inf, 80.0, 30.0):
        for skip_last in (True, False):
            native = globalenergymin._remaining_by_cluster([atoms], adj, max_seed_distance, skip_last=skip_last)
            with mock.patch.object(globalenergymin, '_count_growth_native', lambda *a, **k: None):
                enumerated = globalenergymin._remaining_by_cluster([atoms], adj, max_seed_distance, skip_last=skip_last)
            assert native == enumerated
    total = sum(globalenergymin._remaining_by_cluster([atoms], adj, np.inf, skip_last=True).values())
    if total > 1:
        with pytest.raises(ValueError):
            globalenergymin._remaining_by_cluster([atoms], adj, np.inf, max_amount=total - 1, skip_last=True)
        globalenergymin._remaining_by_cluster([atoms], adj, np.inf, max_amount=total, skip_last=True)


def test_integral_box_slides_at_the_top_left_border():
    """scikit-image's ``_integ`` (used by the determinant-of-Hessian blobs, automation.py:13-38) clips the window's origin first and
    measures the far corner from the CLIPPED origin: a window that starts above / left of the image keeps its size and slides in.
    Hand-computed on a 6x6 image of ones (integral image ii[r, c] = (r + 1)(c + 1)); parity with scikit-image itself is unpinned
    (not importable here)."""
    ii = np.ones((6, 6)).cumsum(0).cumsum(1)
    r = np.array([-2, 0, 3, 5])
    c = np.array([-1, 2])
    got = automation._integ(ii, r, c, 2, 3)
    # rows: origin clipped to (0, 0, 3, 5), far row min(origin + 2, 5) = (2, 2, 5, 5) -> heights 2, 2, 2, 0
    # columns: origin (0, 2), far column min(origin + 3, 5) = (3, 5) -> widths 3, 3
    want = np.array([[2 * 3, 2 * 3], [2 * 3, 2 * 3], [2 * 3, 2 * 3], [0, 0]], float)
    assert np.array_equal(got, want)
    # the unclipped-origin variant would shrink the first window to 0 rows (far row = -2 + 2 = 0)
    assert got[0, 0] == 6.0


def test_lock_step_hands_a_failed_candidate_to_its_own_image_only(monkeypatch):
    """GlobalEnergyMinimization.process_many solves generation k of all images as one batch; a CvxprogError of one image's candidate
    (objects.py:309-318, 351-353) fails THAT image -- the reference fails one image, not the set -- and the other image threads go on."""
    import threading
    from superdsm_amd import globalenergymin as gem

    def fake_multi(jobs, cfg, logs=None, status_line=None, out=None, while_waiting=None, errors=None):
        for j, (objs, y, atoms) in enumerate(jobs):
            for o in objs:
                o.energy = 1.0
            if y == 'bad':
                errors[j] = objects.CvxprogError('convex programming failed', cidx=0)
        return [0] * len(jobs)

    monkeypatch.setattr(gem, 'compute_objects_multi', fake_multi)
    lock = gem._LockStep(3, None)
    res = {}

    def work(name, y, rounds):
        try:
            for _ in range(rounds):
                o = objects.Object()
                o.footprint = {1}
                lock.submit([o], y, None, {}, None)
            res[name] = 'ok'
        except objects.CvxprogError as e:
            res[name] = e.cidx
        finally:
            lock.leave()

    threads = [threading.Thread(target=work, args=('a', 'good', 3)), threading.Thread(target=work, args=('b', 'bad', 3)), threading.Thread(target=work, args=('c', 'good', 2))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=20)
    assert res == {'a': 'ok', 'b': 0, 'c': 'ok'} and lock.batches == 3


def test_cover_update_solves_all_touched_clusters_in_one_native_call():
    """MinSetCover.update (minsetcover.py:142-153) through sdsm_minsetcover_multi: the same solutions as one solve_minsetcover per cluster."""
    rng = np.random.default_rng(3)

    class Adj:
        cluster_labels = [1, 2, 3]

        def get_cluster_label(self, a):
            return 1 + (a - 1) // 5

        def get_atoms_in_cluster(self, cl):
            return set(range(5 * (cl - 1) + 1, 5 * cl + 1))

    def obj(fp, e):
        o = objects.Object()
        o.footprint, o.energy = set(fp), float(e)
        return o

    atoms = [obj({a}, rng.uniform(20, 60)) for a in range(1, 16)]
    cover = minsetcover.MinSetCover(atoms, 25.0, Adj())
    for gen in range(3):
        new = []
        for cl in (1, 2, 3):
            members = sorted(Adj().get_atoms_in_cluster(cl))
            for _ in range(4):
                k = int(rng.integers(2, 5))
                fp = rng.choice(members, k, replace=False).tolist()
                new.append(obj(fp, rng.uniform(15, 45) * k))
        cover.update(new)
        for cl in (1, 2, 3):
            ref = minsetcover.solve_minsetcover(cover.objects_by_cluster[cl], 25.0)
            assert [id(o) for o in cover.solution_by_cluster[cl]] == [id(o) for o in ref]


def test_disk_morphology_from_the_definition():
    """`_morph` (the scikit-image calls of the reference's downstream stages, postprocess.py:155,258-262,270,321, render.py:380-384)
    against the DEFINITIONS, not against another SciPy call (the fixtures of postprocess.npz / render.npz were made with the same SciPy
    restatement, so they cannot pin it): disk(r) = {dy^2 + dx^2 <= r^2} of a (2r+1)^2 window (literal footprints for r = 1, 2, 3 as the
    scikit-image documentation prints them); dilation = union of the footprint translated to every set pixel, clipped to the image;
    erosion = the pixels whose translated footprint, as far as it lies inside the image, is all set (the image border does not erode)."""
    from superdsm_amd import _morph
    np.testing.assert_array_equal(_morph.disk(1), [[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    np.testing.assert_array_equal(_morph.disk(2), [[0, 0, 1, 0, 0], [0, 1, 1, 1, 0], [1, 1, 1, 1, 1], [0, 1, 1, 1, 0], [0, 0, 1, 0, 0]])
    np.testing.assert_array_equal(_morph.disk(3), [[0, 0, 0, 1, 0, 0, 0], [0, 1, 1, 1, 1, 1, 0], [0, 1, 1, 1, 1, 1, 0], [1, 1, 1, 1, 1, 1, 1],
                                                   [0, 1, 1, 1, 1, 1, 0], [0, 1, 1, 1, 1, 1, 0], [0, 0, 0, 1, 0, 0, 0]])
    rng = np.random.default_rng(3)
    for shape, density in (((17, 23), 0.08), ((12, 9), 0.5), ((20, 20), 0.9), ((5, 31), 0.3)):
        img = rng.random(shape) < density
        img[0, :3] = True                                   # pixels on the border and in a corner
        img[-1, -1] = True
        for r in (1, 2, 3, 5):
            fp = _morph.disk(r)
            offs = [(dy - r, dx - r) for dy in range(2 * r + 1) for dx in range(2 * r + 1) if fp[dy, dx]]
            dil = np.zeros(shape, bool)
            ero = np.ones(shape, bool)
            for y in range(shape[0]):
                for x in range(shape[1]):
                    for dy, dx in offs:
                        yy, xx = y + dy, x + dx
                        if 0 <= yy < shape[0] and 0 <= xx < shape[1]:
                            if img[y, x]:
                                dil[yy, xx] = True
                            if not img[yy, xx]:
                                ero[y, x] = False
            np.testing.assert_array_equal(_morph.binary_dilation(img, fp), dil)
            np.testing.assert_array_equal(_morph.binary_erosion(img, fp), ero)


def test_dsm_init_maps_to_the_solver_protocol():
    """dsm/init (objects.py:384-393): 'elliptical' solves the elliptical model first; anything else -- None, or a callable whose return value is the
    starting point (handed over by sdsm_plan_set_start) -- starts the DSM solve directly; the approximation knobs of the CPU implementation are checked as the reference checks them."""
    from superdsm_amd import _capi, objects
    assert _capi.make_config(dict(init='elliptical')).init_elliptical == 1 and _capi.make_config({}).init_elliptical == 1
    assert _capi.make_config(dict(init=None)).init_elliptical == 0
    assert _capi.make_config(dict(init=lambda m: np.zeros(6 + m))).init_elliptical == 0
    cfg = objects._clean_cfg(dict(init=len, hessian_sparsity_tol=1e-3, cachesize=1, scale=1000))
    assert cfg['init'] is len and 'hessian_sparsity_tol' not in cfg and 'cachesize' not in cfg
    with pytest.raises(NotImplementedError):
        objects._clean_cfg(dict(sparsity_tol=1e-3))
    with pytest.raises(AssertionError):
        objects._clean_cfg(dict(hessian_sparsity_tol=-1.0))
