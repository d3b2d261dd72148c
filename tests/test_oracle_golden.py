"""The oracle (oracle/sdsm_oracle.c) against the golden vectors produced by RUNNING the reference
(tests/golden/make_golden.py).  CPU only."""
import glob
import json
import os

import numpy as np
import pytest
import scipy.sparse

from oracle import oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def unpack(bits, shape):
    shape = tuple(int(s) for s in shape)
    return np.unpackbits(bits)[:int(np.prod(shape))].reshape(shape).astype(bool)


@pytest.mark.parametrize('key', ['a', 'b', 'c', 'd'])
def test_preprocess(key):
    d = np.load(os.path.join(G, f'preprocess_{key}.npz'))
    cfg = json.loads(str(d['cfg']))
    y = oracle.preprocess(d['g_raw'], cfg.get('sigma1', np.sqrt(2)), cfg.get('sigma2', 40), cfg.get('offset_clip', 3), cfg.get('lower_clip_mean', False))
    # same association order as SciPy's correlate1d and numpy's pairwise std: a few ulp at most
    np.testing.assert_allclose(y, d['y'], rtol=0, atol=1e-14)


def test_region_reference_kat():
    # literal known-answer test of the reference: tests/test_objects.py:52-75
    y = np.array([[-1, -1, -1, -1, -1], [-1, -1, -1, -1, -1], [-1, -1, -1, -1, -1],
                  [-1, +1, -1, -1, -1], [-1, +1, -1, -1, +1], [-1, +1, -1, -1, +1]], float)
    atoms = np.array([[1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 2], [1, 1, 1, 2, 2], [1, 1, 1, 2, 2], [1, 1, 1, 2, 2]])
    expected = np.array([[0, 0, 0, 0, 0], [0, 1, 0, 0, 0], [1, 1, 1, 0, 0], [1, 1, 1, 0, 0], [1, 1, 1, 0, 0], [1, 1, 1, 0, 0]], bool)
    got = oracle.region_mask(y, None, atoms, [1], 2)
    np.testing.assert_array_equal(got, expected)


def test_region_golden():
    d = np.load(os.path.join(G, 'region.npz'))
    shape = tuple(d['shape'])
    y, atoms = d['y'], d['atoms']
    y_mask = unpack(d['y_mask'], shape)
    # candidate-independent part first: EDT(y <= 0) <= 8
    d2 = oracle.edt_sq(y <= 0)
    np.testing.assert_array_equal(d2 <= 64, unpack(d['edt_le_8'], shape))
    for i in range(3):
        for margin in (3, 8):
            got = oracle.region_mask(y, y_mask, atoms, d[f'fp{i}'].tolist(), margin)
            np.testing.assert_array_equal(got, unpack(d[f'fp{i}_m{margin}'], shape))


def test_edt_exact_bruteforce():
    rng = np.random.default_rng(3)
    nz = rng.random((23, 31)) > 0.08
    d2 = oracle.edt_sq(nz)
    zr, zc = np.nonzero(~nz)
    rr, cc = np.mgrid[:23, :31]
    brute = ((rr[..., None] - zr) ** 2 + (cc[..., None] - zc) ** 2).min(axis=-1)
    np.testing.assert_array_equal(d2, brute)


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(G, 'smoothmat_*.npz'))), ids=lambda p: os.path.basename(p)[10:-4])
def test_smooth_matrix(path):
    d = np.load(path)
    mask = unpack(d['mask'], d['mask_shape'])
    sigma, mult, sub = d['params']
    sm = oracle.smooth_matrix(mask, sigma, mult, int(sub))
    N, M = (int(v) for v in d['shape'])
    assert (sm.N, sm.M) == (N, M)
    np.testing.assert_array_equal(oracle.psf(sigma, mult), d['psf'])          # float32 PSF, bit exact
    if M == 0:
        return
    grid = np.zeros(sm.compressed_shape, bool)
    grid[sm.grid_r, sm.grid_c] = True
    np.testing.assert_array_equal(grid, unpack(d['grid'], d['grid_shape']))  # greedy grid, exact
    np.testing.assert_array_equal(sm.indptr, d['indptr'])
    np.testing.assert_array_equal(sm.indices, d['indices'])
    np.testing.assert_array_equal(sm.data, d['data'])                         # float32-exact entries


def _energy_case(d, k):
    cfg = json.loads(str(d['cfg']))
    y, atoms = d['y'], d['atoms']
    fp = d[f'c{k}_fp'].tolist()
    mask = oracle.region_mask(y, None, atoms, fp, cfg['background_margin'])
    deform = bool(d[f'c{k}_deform'])
    J = oracle.Energy(y, mask, cfg['epsilon'], float(d[f'c{k}_alpha']), cfg['smooth_amount'] if deform else np.inf,
                      cfg['gaussian_shape_multiplier'], cfg['smooth_subsample'])
    return J


def test_energy_value_grad_hessian():
    d = np.load(os.path.join(G, 'energy.npz'))
    n_guard = 0
    for k in range(int(d['n_cases'])):
        J = _energy_case(d, k)
        p = d[f'c{k}_params']
        assert J.n == p.size
        v, g, H = J.eval(p)
        n_guard += int(d[f'c{k}_n_guarded'])
        np.testing.assert_allclose(v, float(d[f'c{k}_value']), rtol=1e-12)
        ref_g = d[f'c{k}_grad']
        np.testing.assert_allclose(g, ref_g, rtol=1e-10, atol=1e-10 * np.abs(ref_g).max())
        ref_H = d[f'c{k}_hessian_lower']
        np.testing.assert_allclose(np.tril(H), ref_H, rtol=1e-10, atol=1e-10 * np.abs(ref_H).max())
        np.testing.assert_allclose(H, H.T, rtol=1e-13, atol=0)
    assert n_guard > 0, 'the exp() guard path (dsm.py:298-300) must be exercised by at least one case'


@pytest.mark.parametrize('tag', ['bbbc039_params', 'large_sigma', 'large_systems'])
def test_tight_optima_and_mask_tail(tag):
    d = np.load(os.path.join(G, f'optimum_{tag}.npz'))
    cfg = json.loads(str(d['cfg']))
    y, atoms = d['y'], d['atoms']
    ncase = int(d['n_cases'])
    fps = [d[f'c{k}_fp'].tolist() for k in range(ncase)]
    recs, frags, params = oracle.compute_objects(y, None, atoms, fps, dict(cfg, init='elliptical'), nthreads=4)
    for k in range(ncase):
        N, M = int(d[f'c{k}_N']), int(d[f'c{k}_M'])
        assert (recs['N'][k], recs['M'][k]) == (N, M)
        mask = oracle.region_mask(y, None, atoms, fps[k], cfg['background_margin'])
        np.testing.assert_array_equal(mask, unpack(d[f'c{k}_region'], y.shape))
        psi_ref = float(d[f'c{k}_psi_dsm'])
        tight = float(d[f'c{k}_gnorm_dsm']) < 1e-8 and float(d[f'c{k}_gnorm_ell']) < 1e-8
        tol = 1e-6 * N / 1000 + 1e-5 * abs(psi_ref)
        if tight:
            assert abs(recs['energy'][k] - psi_ref) <= tol, (k, recs['energy'][k], psi_ref)
            assert recs['status'][k] == 0 and recs['is_optimal'][k] == 1
            # mask: Dice >= 0.999 against the reference's own tail applied to the tight optimum
            ref_frag = unpack(d[f'c{k}_fg_fragment'], d[f'c{k}_fg_shape'])
            full_ref = np.zeros(y.shape, bool)
            o = d[f'c{k}_fg_offset']
            full_ref[o[0]:o[0] + ref_frag.shape[0], o[1]:o[1] + ref_frag.shape[1]] = ref_frag
            full_got = np.zeros(y.shape, bool)
            o2 = recs['fg_offset'][k]
            full_got[o2[0]:o2[0] + frags[k].shape[0], o2[1]:o2[1] + frags[k].shape[1]] = frags[k]
            dice = 2 * (full_ref & full_got).sum() / max(1, full_ref.sum() + full_got.sum())
            assert dice >= 0.999, (k, dice)
            assert bool(recs['on_boundary'][k]) == bool(d[f'c{k}_on_boundary'])
        else:
            # near-separable region: no finite minimiser, psi -> inf psi; the value depends on the stopping rule
            assert recs['energy'][k] <= psi_ref + tol
        # moment-based initialisation (objects.py:287-296)
        np.testing.assert_allclose(oracle.moment_init(y, mask), d[f'c{k}_moment_init'], rtol=1e-9)


def test_postprocess_oracle_matches_reference_fixtures():
    """Contrast response, mask refinement and glare test of the reference (postprocess.py:254-337) on 6 objects."""
    from oracle import postprocess_oracle as po
    d = np.load(os.path.join(G, 'postprocess.npz'))
    g = d['g']
    bg = np.unpackbits(d['background_mask'])[:g.size].reshape(g.shape).astype(bool)
    objs = [(d[f'o{k}_offset'], d[f'o{k}_fragment'].astype(bool)) for k in range(int(d['n']))]
    np.testing.assert_array_equal(po.background_mask(g.shape, objs, 5), bg)
    gs = scipy_gauss(g, 3)
    for k, (off, frag) in enumerate(objs):
        np.testing.assert_allclose(po.compute_contrast(off, frag, g, 5, 5, 1e-4, bg), float(d[f'o{k}_contrast']), rtol=1e-12)
        np.testing.assert_allclose(po.compute_contrast(off, frag, g, 3, 2, 1e-4, bg), float(d[f'o{k}_contrast_b']), rtol=1e-12)
        for tag, (dist, amp, fill) in dict(a=(1, 2, True), b=(2, 1.5, False), c=(0, 2, True)).items():
            o2, f2 = po.process_mask(off, frag, gs, dist, amp, fill)
            np.testing.assert_array_equal(o2, d[f'o{k}_mask_{tag}_offset'])
            np.testing.assert_array_equal(f2, d[f'o{k}_mask_{tag}_fragment'].astype(bool))
        assert int(po.is_glare(off, frag, gs, 0.5, 5)) == int(d[f'o{k}_is_glare'])


def scipy_gauss(g, sigma):
    import scipy.ndimage as ndi
    return ndi.gaussian_filter(g, sigma)


def test_label_maps_match_reference_fixtures():
    """rasterize_labels (render.py:388-451) on disjoint, eroded / dilated, merged and exactly coinciding objects, and the
    regression rows of tests/regression/validate.py:31-36 on the result."""
    from superdsm_amd import objects, render
    d = np.load(os.path.join(G, 'render.npz'))
    shape = tuple(int(v) for v in d['shape'])

    class Obj(objects.BaseObject):
        def __init__(self, off, frag):
            self.fg_offset, self.fg_fragment = np.asarray(off), np.asarray(frag, bool)

    objs = [Obj(d[f'o{k}_offset'], d[f'o{k}_fragment']) for k in range(int(d['n']))]
    data = {'g_raw': np.zeros(shape)}
    for name, kw in dict(plain={}, eroded=dict(dilate=-1), dilated=dict(dilate=1), merged=dict(merge_overlap_threshold=0.5)).items():
        got = render.rasterize_labels(data, objs, **kw)
        np.testing.assert_array_equal(got, d['lab_' + name])
    lab = render.rasterize_labels(data, objs)
    rows = render.label_map_rows(lab)
    assert len(rows) == len(np.unique(lab)) - 1 and all(isinstance(v, str) for row in rows for v in row)
    import scipy.ndimage as ndi
    for area, cx, cy in rows:
        l = lab[int(round(float(cy))), int(round(float(cx)))]
        assert l > 0 and int(area) == (lab == l).sum()
    missing, spurious = render.compare_rows(rows, rows[1:] + [('1', '0.0', '0.0')])
    assert spurious == {rows[0]} and missing == {('1', '0.0', '0.0')}
    # overlapping objects: the overlap goes to one of the two, nothing is lost, wrap-around of a negative background label
    a, b = Obj((2, 2), np.ones((10, 12), bool)), Obj((6, 8), np.ones((12, 10), bool))
    lab2 = render.rasterize_labels(data, [a, b])
    assert set(np.unique(lab2).tolist()) == {0, 1, 2} and (lab2 > 0).sum() == 120 + 120 - 36
    assert render.rasterize_labels(data, [a], background_label=-1)[0, 0] == 65535


def test_solver_approximations_do_not_move_the_optimum():
    """The solver's Hessian is approximate (G~ row entries below 10 % of the row maximum dropped, regulariser curvature blended
    towards its majoriser); psi and its gradient are exact, so the optimum must be the one plain Newton on the reference's EXACT
    Hessian finds.  Both modes on every candidate of the 256x256 scene: same status, energies within the stated tolerance, and
    the approximate mode needs fewer passes."""
    from superdsm_amd import testing
    scene = testing.make_scene('synthetic256', max_size=3)
    args = (scene['y'], None, scene['atoms'], scene['footprints'], scene['dsm_cfg'])
    approx, _, _ = oracle.compute_objects(*args, nthreads=8)
    oracle.set_exact_hessian(True)
    try:
        exact, _, _ = oracle.compute_objects(*args, nthreads=8)
    finally:
        oracle.set_exact_hessian(False)
    np.testing.assert_array_equal(approx['status'], exact['status'])
    for k in range(len(approx)):
        tol = 1e-6 * exact['N'][k] / 1000 + 1e-5 * abs(exact['energy'][k])
        assert abs(approx['energy'][k] - exact['energy'][k]) <= tol, (k, approx['energy'][k], exact['energy'][k])
    assert approx['evals'].sum() < exact['evals'].sum()
