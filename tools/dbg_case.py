import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, golden_util
from contrack_amd import _native
name = sys.argv[1]
g = golden_util.load(name)
trk = _native.Tracker(0)
f, n = trk.track(g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
print("stats", trk.stats())
d = f != g["flag"]
print("ndiff", d.sum(), "n", n, len(np.unique(g["flag"])) - 1)
tt, yy, xx = np.nonzero(d)
for lab_g in np.unique(g["flag"][d])[:10]:
    m = d & (g["flag"] == lab_g)
    print("golden", lab_g, "-> got", np.unique(f[m]), "t", np.unique(np.nonzero(m)[0])[:8], "y", np.nonzero(m)[1].min(), np.nonzero(m)[1].max(), "x", np.nonzero(m)[2].min(), np.nonzero(m)[2].max())
