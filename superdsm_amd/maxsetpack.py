"""Greedy max-weight set packing on the host (reference: superdsm/maxsetpack.py:8-24), restated."""
from .output import get_output


def solve_maxsetpack(objects, out=None):
    """Repeatedly takes the highest-energy object and drops everything that overlaps it."""
    out = get_output(out)
    pool = list(objects)
    packed = []
    while pool:
        top = max(pool, key=lambda c: c.energy)
        packed.append(top)
        pool = [c for c in pool if not (c.footprint & top.footprint)]
    out.write(f'MAXSETPACK - GREEDY accepted objects: {len(packed)}')
    return packed
