"""scipy.ndimage / numpy restatement of the reference CPU path, for the cpu_baseline leg of bench.py.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (never imported by contrack_amd/).  The reference's Python does not
travel to the GPU box, so this module performs the same library-call sequence on a (time, lat, lon) numpy
slab as contrack/contrack.py:646-796 does -- int64 `where`, two `ndimage.label` calls with the reference's
two 3x3x3 structures, the per-row seam loops, per-step `ndimage.find_objects` with the walk over ALL label
slots (the O(T x labels) term that makes the reference super-linear in T), masked float64 `np.sum`s,
bbox-confined relabels, persistence -- so that its wall time is representative of the reference on the same
host.  Validated against the imported reference (identical flag arrays; tests/test_oracle_golden.py).
One core: none of these calls thread.
"""
import numpy as np
from scipy import ndimage

_PLANE = np.zeros((3, 3, 3), dtype=int)
_PLANE[1] = 1                                   # 8-connectivity inside a timestep only   (contrack.py:684-686)
_TRACK = _PLANE.copy()
_TRACK[0, 1, 1] = _TRACK[2, 1, 1] = 1           # ... plus the same pixel one step back / ahead (contrack.py:748-750)

_COMPARE = {">=": np.greater_equal, "ge": np.greater_equal, "<=": np.less_equal, "le": np.less_equal,
            ">": np.greater, "gt": np.greater, "<": np.less, "lt": np.less}


def _seam_rows(lab):
    """yield (t, y) in scan order for rows whose first and last pixel are both labelled"""
    first, last = lab[:, :, 0], lab[:, :, -1]
    for t in range(lab.shape[0]):
        for y in range(lab.shape[1]):
            if first[t, y] > 0 and last[t, y] > 0:
                yield t, y


def run_contrack(anom, threshold, gorl, wrow, overlap, persistence, twosided=True):
    with np.errstate(divide="ignore", invalid="ignore"):
        return _run(anom, threshold, gorl, wrow, overlap, persistence, twosided)


def _run(anom, threshold, gorl, wrow, overlap, persistence, twosided):
    if gorl not in _COMPARE:
        raise ValueError(' Please select from [>, >=, <, >=] for gorl')
    T, ny, nx = anom.shape
    binary = np.where(_COMPARE[gorl](anom, threshold), 1, 0)                    # int64, contrack.py:665
    lab, _ = ndimage.label(binary, structure=_PLANE)                            # contrack.py:684
    for t in range(T):                                                          # contrack.py:691-698
        plane = lab[t]
        for y in range(ny):
            a, b = plane[y, 0], plane[y, -1]
            if a > 0 and b > 0 and a != b:
                plane[plane == max(a, b)] = min(a, b)
    wgrid = np.ones((ny, nx)) * np.asarray(wrow, dtype=np.float32)[:, None]     # contrack.py:704
    for t in range(1, T - 1):                                                   # contrack.py:706-742
        cur, nxt, prv = lab[t], lab[t + 1], lab[t - 1]
        for slot, box in enumerate(ndimage.find_objects(cur)):
            if box is None:
                continue
            ident = slot + 1
            inside = cur[box] == ident
            wbox = wgrid[box]
            area = np.sum(wbox[inside])
            fwd = np.sum(wbox[inside & (nxt[box] >= 1)])
            bwd = np.sum(wbox[inside & (prv[box] >= 1)])
            fb = (1 / area) * bwd
            ff = (1 / area) * fwd
            if twosided:
                drop = ((fb != 0 and ff != 0 and (fb < overlap or ff < overlap)) or
                        (fb != 0 and ff == 0 and fb < overlap) or
                        (fb == 0 and ff != 0 and ff < overlap))
            else:
                drop = ff < overlap
            if drop:
                cur[box][inside] = 0
    binary = np.where(lab >= 1, 1, 0)                                           # contrack.py:747
    lab, _ = ndimage.label(binary, structure=_TRACK)                            # contrack.py:748
    boxes = ndimage.find_objects(lab)                                           # contrack.py:753 (once)
    for t, y in _seam_rows(lab):                                                # contrack.py:754-763
        a, b = lab[t, y, 0], lab[t, y, -1]
        if a > 0 and b > 0 and a != b:
            hi, lo = max(a, b), min(a, b)
            region = lab[boxes[hi - 1]]
            region[region == hi] = lo
    for slot, box in enumerate(ndimage.find_objects(lab)):                      # contrack.py:765-772
        if box is not None and (box[0].stop - box[0].start) < persistence:
            region = lab[box]
            region[region == slot + 1] = 0
    return lab, len(np.unique(lab)) - 1                                         # contrack.py:793
