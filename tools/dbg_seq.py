import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, golden_util
from contrack_amd import _native
trk = _native.Tracker(0)
prev = None
for name in golden_util.case_names():
    g = golden_util.load(name)
    f, n = trk.track(g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
    st = trk.stats()
    ok = np.array_equal(f, g["flag"])
    if not ok:
        f2, n2 = trk.track(g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
        print(name, "FAIL after", prev, "shape", g["anom"].shape, "ndiff", (f != g["flag"]).sum(), "rerun ok:", np.array_equal(f2, g["flag"]),
              {k: st[k] for k in ("seam_rows_to_driver", "labels_3d", "seam_ops", "filter_passes")})
    prev = name
print("done")
