"""Round 5: the time of k_threshold (and k_relabel) on eight handles of one process, on the same slab -- the "modes" of the kernel per
handle.  CTK_MASK_TUNE=0: without the mask placement check; CTK_THR_STORE=2: the kernel without its stores; SHAPE=T,ny,nx; NH handles.
(tools/mask_check_probe.sh, tools/mask_store_probe.sh)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrack_amd import _native
T, ny, nx = [int(x) for x in os.environ.get("SHAPE", "2707,181,360").split(",")]
n = T * ny * nx * 4
lat = np.linspace(90, -90, ny).astype(np.float32)
w = (111 * 111 * np.cos(lat * np.pi / 180)).astype(np.float32)
thr = np.full(T, 160.0)
t0 = _native.Tracker(0)
d_in, d_out = t0.malloc(n), t0.malloc(n)
t0.synth_fill(d_in, T, ny, nx, seed=0)
keep = []
out = []
for k in range(int(os.environ.get('NH', '8'))):
    trk = _native.Tracker(0)
    trk.set_timing(1)
    for _ in range(3):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
    trk.timing_sums(reset=True)
    for _ in range(16):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
    per, _ = trk.timing_sums(reset=True)
    out.append((per["k_threshold"], per["k_relabel"]))
    keep.append(trk)
print("tune=%s  thr: %s   rel: %s" % (os.environ.get("CTK_MASK_TUNE", "1"),
                                                  " ".join("%.4f" % a for a, _ in out), " ".join("%.4f" % b for _, b in out)))
