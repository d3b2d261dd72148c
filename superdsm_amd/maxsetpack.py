"""Greedy max-weight set packing on the host (reference: superdsm/maxsetpack.py:8-24), restated."""
from .output import get_output


def solve_maxsetpack(objects, out=None):
    """Repeatedly takes the highest-energy object and drops everything that overlaps it (native host code, sdsm_maxsetpack;
    :func:`solve_maxsetpack_py` is the restatement it is tested against)."""
    import ctypes
    import numpy as np
    from . import _capi
    from .minsetcover import _bitsets
    objects = list(objects)
    if len(objects) < 4:
        return solve_maxsetpack_py(objects, out)
    energies = np.ascontiguousarray([c.energy for c in objects], np.float64)
    if not np.isfinite(energies).all():
        return solve_maxsetpack_py(objects, out)
    masks, words = _bitsets(objects)
    sel = np.zeros(len(objects), np.int32)
    nsel = ctypes.c_int32(0)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    code = _capi.lib().sdsm_maxsetpack(len(objects), words, ptr(masks), ptr(energies), ptr(sel), ctypes.byref(nsel))
    assert code == 0
    get_output(out).write(f'MAXSETPACK - GREEDY accepted objects: {nsel.value}')
    return [objects[i] for i in sel[:nsel.value]]


def solve_maxsetpack_py(objects, out=None):
    """The same in Python, statement by statement after the reference."""
    out = get_output(out)
    pool = list(objects)
    packed = []
    while pool:
        top = max(pool, key=lambda c: c.energy)
        packed.append(top)
        pool = [c for c in pool if not (c.footprint & top.footprint)]
    out.write(f'MAXSETPACK - GREEDY accepted objects: {len(packed)}')
    return packed
