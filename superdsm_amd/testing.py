"""Scene construction and one-call GPU solves shared by tests, __graft_entry__.smoke() and bench.py."""
import numpy as np

from . import synth
from .atoms import AtomAdjacencyGraph


def make_scene(workload='synthetic256', max_size=3, alpha_factor=None, layout_index=0):
    """Synthetic image -> y, atoms, adjacency graph, candidate footprints and the dsm/* hyper-parameters of the
    BASELINE.json config the workload stands for.  ``layout_index`` (bbbc039_like only): which of the
    eight reference object tables places the nuclei."""
    spec = dict(synth.WORKLOADS[workload])
    if workload == 'bbbc039_like':
        shape, layout = synth.bbbc039_like_layout(spec['seed'], layout_index)
        spec['seed'] += 7919 * layout_index
        af = 0.00033 if alpha_factor is None else alpha_factor       # examples/BBBC039/task.json: AF_alpha
    else:
        shape = spec['shape']
        layout = synth.random_layout(shape, spec['n'], spec['radius'], spec['seed'], min_sep={'synthetic4096': 0.6, 'synthetic512': 1.2}.get(workload, 2.2))
        af = {'synthetic256': 0.00033, 'synthetic512': 0.00033, 'synthetic4096': 0.00033, 'gowt1_like': 0.0005, 'nih3t3_like': 0.000375}[workload] if alpha_factor is None else alpha_factor
    g = synth.render_image(shape, layout, spec['seed'])
    y = synth.offset_image(g, spec['scale'])
    atoms, clusters, seeds = synth.make_atoms(y, layout, spec['seed'])
    adj = AtomAdjacencyGraph(atoms, clusters, y > 0, seeds)
    footprints = synth.enumerate_candidates(adj, max_size=max_size)
    return dict(workload=workload, g=g, y=y, atoms=atoms, clusters=clusters, seeds=seeds, adjacencies=adj, footprints=footprints,
                dsm_cfg=synth.dsm_config_for_scale(spec['scale'], af), scale=spec['scale'])


def solve_scene_gpu(scene, footprints=None, want_xi=False):
    from . import engine
    import torch
    fps = scene['footprints'] if footprints is None else footprints
    img = engine.DeviceImage(scene['y'], None, scene['atoms'], scene['dsm_cfg']['background_margin'])
    batch = engine.Batch(img, fps, scene['dsm_cfg'], want_xi=want_xi)
    batch.launch()
    torch.cuda.synchronize()
    recs = batch.records()
    frags = batch.fragments(recs)
    out = dict(records=recs, fragments=frags, batch=batch, image=img)
    if want_xi:
        out['xi'] = batch.xi_dev.cpu().numpy()
        out['xi_offsets'] = batch.xi_offsets()
    return out


def dice(a_off, a_frag, b_off, b_frag, shape):
    fa = np.zeros(shape, bool)
    fb = np.zeros(shape, bool)
    fa[a_off[0]:a_off[0] + a_frag.shape[0], a_off[1]:a_off[1] + a_frag.shape[1]] = a_frag
    fb[b_off[0]:b_off[0] + b_frag.shape[0], b_off[1]:b_off[1] + b_frag.shape[1]] = b_frag
    den = fa.sum() + fb.sum()
    return 1.0 if den == 0 else 2.0 * (fa & fb).sum() / den
