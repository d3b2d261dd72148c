"""Label maps of segmentation results and the reference's regression metric.

``rasterize_labels`` restates superdsm/render.py:388-451 (objects -> uniquely labelled uint16 image: optional merging of
strongly overlapping objects, overlapping pixels handed to the nearest object by a marker-based watershed on the distance map,
exactly coinciding objects kept).  ``label_map_rows`` / ``compare_rows`` restate tests/regression/validate.py:31-36,76-80: a
label map is summarised as the SET of rows (area, round(centre x, 1), round(centre y, 1)) -- strings -- and two results agree iff
the sets are equal; this is how the reference's committed ``tests/regression/expected/<host>/<task>/<image>.csv`` files are
compared.  Host code (the reference's own is NumPy / scikit-image on the host too); the watershed of scikit-image is replaced by
a priority flood with its documented semantics (4-connectivity, ties by insertion order)."""
import csv
import heapq

import numpy as np
import scipy.ndimage as ndi

from . import _morph


def render_objects_foregrounds(shape, objects):
    """One full-image bool mask per object, in turn (superdsm/_aux.py:51-56)."""
    for obj in objects:
        foreground = np.zeros(shape, bool)
        obj.fill_foreground(foreground)
        yield foreground


def rasterize_objects(data, objects, dilate=0):
    """Yields the (optionally dilated / eroded) masks of the objects that have any foreground (render.py:368-385)."""
    if isinstance(objects, str):
        objects = list(data[objects])
    for foreground in render_objects_foregrounds(data['g_raw'].shape, objects):
        if dilate > 0:
            foreground = _morph.binary_dilation(foreground, _morph.disk(dilate))
        elif dilate < 0:
            foreground = _morph.binary_erosion(foreground, _morph.disk(-dilate))
        if foreground.any():
            yield foreground.copy()


def _watershed(image, markers, mask):
    """Marker-based watershed by priority flood (what ``skimage.segmentation.watershed(image, markers, mask=mask)`` computes with
    its defaults): pixels enter a heap keyed by (image value, insertion order); a popped pixel gives its label to its unlabelled
    4-neighbours inside ``mask``, which enter the heap in turn.  Only marker pixels with an unlabelled neighbour can ever label
    anything, so only those are queued (in raster order, which keeps the relative insertion order of the full algorithm)."""
    out = np.where(mask, markers, 0).astype(np.int64)
    todo = mask & (out == 0)
    if not todo.any():
        return out
    H, W = out.shape
    near = ndi.binary_dilation(todo, structure=np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool)) & (out > 0)
    heap = []
    age = 0
    for r, c in zip(*np.nonzero(near)):
        heap.append((float(image[r, c]), age, int(r), int(c)))
        age += 1
    heapq.heapify(heap)
    while heap:
        _, _, r, c = heapq.heappop(heap)
        lab = out[r, c]
        for rr, cc in ((r - 1, c), (r + 1, c), (r, c - 1), (r, c + 1)):
            if 0 <= rr < H and 0 <= cc < W and out[rr, cc] == 0 and mask[rr, cc]:
                out[rr, cc] = lab
                heapq.heappush(heap, (float(image[rr, cc]), age, rr, cc))
                age += 1
    return out


def rasterize_labels(data, objects='postprocessed_objects', merge_overlap_threshold=np.inf, dilate=0, background_label=0):
    """Integer image of uniquely labelled segmentation masks (render.py:388-451).

    :param data: pipeline data (``g_raw`` gives the shape).
    :param objects: name of the output to rasterise, or a list of objects with ``fill_foreground``.
    :param merge_overlap_threshold: pairs overlapping by more than this fraction of the smaller one are merged.
    :param dilate: dilate (> 0) or erode (< 0) every mask by a disk of this radius first.
    :param background_label: label of the background (non-positive).
    """
    assert background_label <= 0
    objects = list(rasterize_objects(data, objects, dilate))

    merge_list = []
    if merge_overlap_threshold <= 1:
        for i1 in range(len(objects)):
            for i2 in range(i1):
                overlap = np.logical_and(objects[i1], objects[i2]).sum() / (0. + min(objects[i1].sum(), objects[i2].sum()))
                if overlap > merge_overlap_threshold:
                    merge_list.append((i1, i2))
    labels = list(range(1, 1 + len(objects)))
    members = {label: [label - 1] for label in labels}
    for merge_idx, (i1, i2) in enumerate(merge_list):
        new_label = len(objects) + 1 + merge_idx
        l1, l2 = labels[i1], labels[i2]
        if l1 == l2:
            continue                                         # already merged (transitivity)
        merged = members[l1] + members[l2]
        for k in merged:
            labels[k] = new_label
        members[new_label] = merged
        del members[l1], members[l2]
    objects = [np.sum([objects[k] for k in ks], axis=0) > 0 for ks in members.values()]

    result = np.zeros(data['g_raw'].shape, 'uint16')
    if len(objects) > 0:
        overlaps = np.sum(objects, axis=0) > 1
        for l, obj in enumerate(objects, 1):
            result[obj] = l
        background = result == 0
        result[overlaps] = 0
        dist = ndi.distance_transform_edt(result == 0)
        result = _watershed(dist, result, ~background)
        assert not (result < 0).any() and not (result >= 2 ** 16).any()
        result = result.astype('uint16')
    # two or more objects that coincide exactly have been eliminated by the steps above: give them a label each (render.py:443-447)
    for obj in objects:
        lost = ((result > 0) * 1 - (obj > 0) * 1 < 0)
        if lost.any():
            result[lost] = result.max() + 1
    # (the reference assigns the non-positive background label into its uint16 image: under the NumPy it pins, 1.20, a negative
    #  value wraps -- background_label = -1 reads 65535 --, kept here)
    result[result == 0] = np.array(background_label).astype('uint16')
    return result


# ---- regression metric (tests/regression/validate.py) -----------------------------------------------------------------
def label_map_rows(labels):
    """The rows of validate.py:31-36 for one label map: per label ``(str(area), str(round(centre_x, 1)), str(round(centre_y, 1)))``,
    sorted by the centre columns as the CSV writer does (validate.py:38)."""
    labels = np.asarray(labels)
    rows = []
    for l in sorted(frozenset(labels.reshape(-1).tolist()) - {0}):
        cc = labels == l
        cy, cx = ndi.center_of_mass(cc)
        rows.append((str(int(cc.sum())), str(round(cx, 1)), str(round(cy, 1))))
    rows.sort(key=lambda row: row[1:3])
    return rows


def write_rows_csv(path, rows):
    with open(path, 'w', newline='') as fp:
        csv.writer(fp, delimiter=',', quoting=csv.QUOTE_ALL).writerows([['Object size', 'Center X', 'Center Y']] + list(rows))


def read_rows_csv(path):
    with open(path, newline='') as fp:
        return [tuple(row) for k, row in enumerate(csv.reader(fp, delimiter=',', quoting=csv.QUOTE_ALL)) if k > 0]


def compare_rows(actual_rows, expected_rows):
    """validate.py:76-80: (missing, spurious) as sets; the results agree iff both are empty."""
    a, e = frozenset(map(tuple, actual_rows)), frozenset(map(tuple, expected_rows))
    return e - a, a - e


def regression_agreement(actual_rows, expected_rows):
    """Fraction of the expected objects that are matched exactly, and the two mismatch counts."""
    missing, spurious = compare_rows(actual_rows, expected_rows)
    n = max(1, len(frozenset(map(tuple, expected_rows))))
    return dict(expected=len(frozenset(map(tuple, expected_rows))), missing=len(missing), spurious=len(spurious), matched_fraction=1 - len(missing) / n)
