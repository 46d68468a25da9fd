"""Is k_threshold's per-process mode a matter of the HANDLE's work space (mask, thresholds)?  Several handles in one process,
dummy allocations between them, the same slab."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrack_amd import _native
T, ny, nx = 2707, 181, 360
n = T * ny * nx * 4
lat = np.linspace(90, -90, ny).astype(np.float32)
w = (111 * 111 * np.cos(lat * np.pi / 180)).astype(np.float32)
thr = np.full(T, 160.0)
t0 = _native.Tracker(0)
d_in, d_out = t0.malloc(n), t0.malloc(n)
t0.synth_fill(d_in, T, ny, nx, seed=0)
keep = []
sizes = [0, 8, 15, 22, 29, 36, 1, 64, 3, 100, 17, 256, 5, 33, 2, 512]
for k in range(int(os.environ.get('NH', '6'))):
    if sizes[k]:
        keep.append(t0.malloc(sizes[k] << 20))
    trk = _native.Tracker(0)
    trk.set_timing(1)
    for _ in range(3):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
    trk.timing_sums(reset=True)
    for _ in range(24):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
    per, _ = trk.timing_sums(reset=True)
    print("handle %d  thr %.4f  rel %.4f" % (k, per["k_threshold"], per["k_relabel"]))
    keep.append(trk)
