"""run_lifecycle reductions on random flag planes / fields against the scipy port.  python tools/fuzz_lifecycle.py [first] [count]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from contrack_amd import _native
from contrack_amd.contrack import lifecycle_frame, row_weights
from oracle import lifecycle_port
from scipy import ndimage

first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 200)
bad = []
with _native.Tracker(0) as trk:
    for i in range(first, first + count):
        rng = np.random.default_rng(90000 + i)
        T = int(rng.integers(1, 7)); ny = int(rng.integers(3, 40)); nx = int(rng.choice([4, 7, 16, 33, 64, 65, 100, 130]))
        # labelled blobs: threshold a smooth-ish random field, label with wrap-unaware scipy, then join ids across the seam at random
        f = ndimage.uniform_filter(rng.standard_normal((T, ny, nx)), size=(1, 3, 5), mode=("nearest", "nearest", "wrap"))
        flag = np.zeros((T, ny, nx), np.int32)
        for t in range(T):
            lab, n = ndimage.label(f[t] > 0.15)
            perm = rng.permutation(np.arange(1, n + 1)) * int(rng.choice([1, 1, 7])) if n else np.array([], int)
            flag[t] = np.where(lab > 0, np.concatenate([[0], perm])[lab], 0)
            for y in range(ny):                                  # merge across the seam sometimes
                if flag[t, y, 0] and flag[t, y, -1] and rng.random() < 0.7:
                    flag[t][flag[t] == flag[t, y, -1]] = flag[t, y, 0]
        if rng.random() < 0.2:
            flag[flag == flag.max()] = -5                            # a negative id
        f64 = bool(rng.integers(0, 2))
        field = (rng.random((T, ny, nx)) * 50 + 100).astype(np.float64 if f64 else np.float32)
        lat = np.linspace(90, -90, ny).astype(np.float32); lon = (np.arange(nx) * (360.0 / nx)).astype(np.float32)
        wrow = row_weights(lat, 180.0 / (ny - 1), 360.0 / nx)
        dates = ["%02d" % t for t in range(T)]
        want = lifecycle_port.run_lifecycle(flag, field, lat, lon, wrow, dates)
        rows = trk.lifecycle(flag, field, wrow)
        got = lifecycle_frame(rows, lat, lon, dates, flag, field, wrow)
        ok = len(got) == len(want) and all(a == b
                                          for a, b in zip(got, want))
        if not ok:
            diff = [(a, b) for a, b in zip(got, want) if a != b][:2]
            bad.append((i, (T, ny, nx), len(got), len(want), diff))
print("lifecycle fuzz %d..%d: problems %d %s" % (first, first + count - 1, len(bad), bad[:3]))
