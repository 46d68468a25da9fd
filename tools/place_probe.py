"""Which work-space buffer's placement moves k_relabel (and k_threshold)?  One handle; one buffer at a time is dropped and allocated
anew (ctk_debug_drop_buffer), several times in a row; the kernel times after each re-allocation."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrack_amd import _native
T, ny, nx = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2707, 181, 360)
n = T * ny * nx * 4
lat = np.linspace(90, -90, ny).astype(np.float32)
w = (111 * 111 * np.cos(lat * np.pi / 180)).astype(np.float32)
thr = np.full(T, 160.0)
trk = _native.Tracker(0)
d_in, d_out = trk.malloc(n), trk.malloc(n)
trk.synth_fill(d_in, T, ny, nx, seed=0)
L = _native.lib()
trk.set_timing(1)
def measure():
    for _ in range(3):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
    trk.timing_sums(reset=True)
    for _ in range(16):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
    per, _ = trk.timing_sums(reset=True)
    return per["k_threshold"], per["k_relabel"]
print("start: thr %.4f rel %.4f" % measure())
names = ["mask", "wstart", "rowstart", "chunk_vals", "run_val", "run_base"]
for which in (1, 2, 3, 4, 5, 0):
    out = []
    for rep in range(6):
        _native.check(L.ctk_debug_drop_buffer(trk.handle, which))
        out.append(measure())
    print("%-10s re-allocated: thr %s | rel %s" % (names[which], " ".join("%.4f" % a for a, _ in out), " ".join("%.4f" % b for _, b in out)))
# and the OUTPUT buffer
out = []
for rep in range(6):
    trk.free(d_out); d_out = trk.malloc(n)
    out.append(measure())
print("%-10s re-allocated: thr %s | rel %s" % ("flag (out)", " ".join("%.4f" % a for a, _ in out), " ".join("%.4f" % b for _, b in out)))
out = []
for rep in range(4):
    trk.free(d_in); d_in = trk.malloc(n); trk.synth_fill(d_in, T, ny, nx, seed=0)
    out.append(measure())
print("%-10s re-allocated: thr %s | rel %s" % ("anom (in)", " ".join("%.4f" % a for a, _ in out), " ".join("%.4f" % b for _, b in out)))
