"""Does the placement of the CALLER's slab move the threshold kernel as the mask's does?  Several allocations of the slab; for each a
plain read stream (the checksum kernel) and the pass's kernel times (the mask tuned by the library)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrack_amd import _native
T, ny, nx = 2707, 181, 360
n = T * ny * nx
lat = np.linspace(90, -90, ny).astype(np.float32)
w = (111 * 111 * np.cos(lat * np.pi / 180)).astype(np.float32)
thr = np.full(T, 160.0)
trk = _native.Tracker(0)
d_out = trk.malloc(n * 4)
trk.set_timing(1)
hold = []
pre = int(os.environ.get("PRE_MB", "0"))
if pre:
    hold.append(trk.malloc(pre << 20))
for k in range(int(os.environ.get("NS", "8"))):
    d_in = trk.malloc(n * 4)
    trk.synth_fill(d_in, T, ny, nx, seed=0)
    trk.sync()
    for _ in range(3):
        trk.checksum_i32(d_in, n)
    t0 = time.perf_counter()
    for _ in range(20):
        trk.checksum_i32(d_in, n)
    rd = (time.perf_counter() - t0) / 20
    for _ in range(3):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
    trk.timing_sums(reset=True)
    for _ in range(16):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
    per, _ = trk.timing_sums(reset=True)
    print("slab %d at %#x: read probe %.4f ms (incl. call) | thr %.4f rel %.4f" % (k, d_in.value, rd * 1e3, per["k_threshold"], per["k_relabel"]), flush=True)
    if os.environ.get("KEEP", "1") == "1":
        hold.append(d_in)
    else:
        trk.free(d_in)
