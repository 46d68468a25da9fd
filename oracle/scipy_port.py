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


def run_contrack(anom, threshold, gorl, wrow, overlap, persistence, twosided=True):
    with np.errstate(divide="ignore", invalid="ignore"):
        return _run(anom, threshold, gorl, wrow, overlap, persistence, twosided)


def _run(anom, threshold, gorl, wrow, overlap, persistence, twosided):
    if gorl not in _COMPARE:
        raise ValueError(' Please select from [>, >=, <, >=] for gorl')
    T, ny, nx = anom.shape
    binary = np.where(_COMPARE[gorl](anom, threshold), 1, 0)                    # int64, contrack.py:665
    lab, _ = ndimage.label(binary, structure=_PLANE)                            # contrack.py:684
    # The loops below evaluate what the reference evaluates, as often as it does (the same scalar look-ups per seam row,
    # the same three mask expressions per contour, two relabel statements per seam row): the wall time is meant to be the
    # reference's, not that of a tidied-up version.
    for t in range(T):                                                          # contrack.py:691-698
        for y in range(ny):
            if lab[t, y, 0] > 0 and lab[t, y, -1] > 0 and (lab[t, y, 0] > lab[t, y, -1]):
                lab[t][lab[t] == lab[t, y, 0]] = lab[t, y, -1]
            if lab[t, y, 0] > 0 and lab[t, y, -1] > 0 and (lab[t, y, 0] < lab[t, y, -1]):
                lab[t][lab[t] == lab[t, y, -1]] = lab[t, y, 0]
    wgrid = np.ones((ny, nx)) * np.asarray(wrow, dtype=np.float32)[:, None]     # contrack.py:704
    for t in range(1, T - 1):                                                   # contrack.py:706-742
        ident = 0
        for box in ndimage.find_objects(lab[t]):
            ident = ident + 1
            if box is None:
                continue
            area = np.sum(wgrid[box][lab[t][box] == ident])
            fwd = np.sum(wgrid[box][(lab[t][box] == ident) & (lab[t + 1][box] >= 1)])
            bwd = np.sum(wgrid[box][(lab[t][box] == ident) & (lab[t - 1][box] >= 1)])
            fb = (1 / area) * bwd
            ff = (1 / area) * fwd
            if twosided:
                if fb != 0 and ff != 0:
                    if (fb < overlap) or (ff < overlap):
                        lab[t][box][(lab[t][box] == ident)] = 0.
                if fb != 0 and ff == 0:
                    if fb < overlap:
                        lab[t][box][(lab[t][box] == ident)] = 0.
                if fb == 0 and ff != 0:
                    if ff < overlap:
                        lab[t][box][(lab[t][box] == ident)] = 0.
            else:
                if ff < overlap:
                    lab[t][box][(lab[t][box] == ident)] = 0.
    binary = np.where(lab >= 1, 1, 0)                                           # contrack.py:747
    lab, _ = ndimage.label(binary, structure=_TRACK)                            # contrack.py:748
    boxes = ndimage.find_objects(lab)                                           # contrack.py:753 (once)
    for t in range(T):                                                          # contrack.py:754-763
        for y in range(ny):
            if lab[t, y, 0] > 0 and lab[t, y, -1] > 0 and (lab[t, y, 0] > lab[t, y, -1]):
                box = boxes[lab[t, y, 0] - 1]
                lab[box][(lab[box] == lab[t, y, 0])] = lab[t, y, -1]
            if lab[t, y, 0] > 0 and lab[t, y, -1] > 0 and (lab[t, y, 0] < lab[t, y, -1]):
                box = boxes[lab[t, y, -1] - 1]
                lab[box][(lab[box] == lab[t, y, -1])] = lab[t, y, 0]
    ident = 0
    for box in ndimage.find_objects(lab):                                       # contrack.py:765-772
        ident = ident + 1
        if box is None:
            continue
        if (box[0].stop - box[0].start) < persistence:
            lab[box][(lab[box] == ident)] = 0.
    return lab, len(np.unique(lab)) - 1                                         # contrack.py:793
