import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, golden_util
from contrack_amd import _native
trk = _native.Tracker(0)
for name in sys.argv[1:]:
    g = golden_util.load(name)
    for k in range(3):
        f, n = trk.track(g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
        st = trk.stats()
        print(name, k, "ok" if np.array_equal(f, g["flag"]) else "FAIL", {kk: st[kk] for kk in ("seam_rows_to_driver", "labels_3d", "seam_ops", "seam_folds")})
