"""run_lifecycle reductions on random flag planes / fields against the scipy port.  python tools/fuzz_lifecycle.py [first] [count]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from contrack_amd import _native
from contrack_amd.contrack import lifecycle_frame, row_weights
from oracle import lifecycle_port
import life_util

first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 200)
bad = []
with _native.Tracker(0) as trk:
    for i in range(first, first + count):
        flag, field, lat, lon, wrow, dates = life_util.random_life_case(i)
        T, ny, nx = flag.shape
        want = lifecycle_port.run_lifecycle(flag, field, lat, lon, wrow, dates)
        rows = trk.lifecycle(flag, field, wrow)
        got = lifecycle_frame(rows, lat, lon, dates, trk)
        ok = len(got) == len(want) and all(a == b
                                          for a, b in zip(got, want))
        if not ok:
            diff = [(a, b) for a, b in zip(got, want) if a != b][:2]
            bad.append((i, (T, ny, nx), len(got), len(want), diff))
print("lifecycle fuzz %d..%d: problems %d %s" % (first, first + count - 1, len(bad), bad[:3]))
