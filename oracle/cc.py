"""ORACLE (test infrastructure only): CPU restatement of the reference's connected-component op and hole filling.

* ``connected_components``: contract of sam2/_C ``get_connected_componnets`` (csrc/connected_components.cu:213-289,
  documented in sam2/utils/misc.py:48-61): 8-connectivity, labels > 0 per component / 0 background, counts = area.
* ``fill_holes_in_mask_scores``: sam2/utils/misc.py:365-393.

**Parity unpinned for the labelling itself**: the reference implementation is CUDA-only (it cannot be built or run in
this container, and the CPU reference silently skips hole filling, misc.py:389-391), so this restates the documented
contract with ``scipy.ndimage.label`` and is cross-checked against a brute-force flood fill in tests/.  Label
VALUES are implementation-defined (the reference numbers components by the first 2x2 block's top-left pixel);
callers only use ``labels > 0`` and ``counts``, which is what tests compare (plus partition equivalence).
"""
import numpy as np
from scipy import ndimage

_EIGHT = np.ones((3, 3), dtype=np.int32)


def connected_components(mask):
    """mask [N,1,H,W] array-like (non-zero = fg) -> (labels int32, counts int32); label = min raster index + 1."""
    m = np.asarray(mask) != 0
    labels = np.zeros(m.shape, dtype=np.int32)
    counts = np.zeros(m.shape, dtype=np.int32)
    N = m.shape[0]
    for i in range(N):
        lab, n = ndimage.label(m[i, 0], structure=_EIGHT)
        if n == 0:
            continue
        area = np.bincount(lab.ravel(), minlength=n + 1)
        flat = lab.ravel()
        idx = np.arange(flat.size)
        first = np.full(n + 1, flat.size, dtype=np.int64)
        np.minimum.at(first, flat, idx)
        fg = lab > 0
        labels[i, 0][fg] = (first[lab[fg]] + 1).astype(np.int32)
        counts[i, 0][fg] = area[lab[fg]].astype(np.int32)
    return labels, counts


def fill_holes_in_mask_scores(mask, max_area):
    """misc.py:365-393 with the extension present: holes = components of (mask <= 0) with area <= max_area -> 0.1."""
    assert max_area > 0
    m = np.asarray(mask, dtype=np.float32)
    shp = m.shape
    m4 = m.reshape(-1, 1, shp[-2], shp[-1])
    labels, areas = connected_components(m4 <= 0)
    out = np.where((labels > 0) & (areas <= max_area), np.float32(0.1), m4)
    return out.reshape(shp)
