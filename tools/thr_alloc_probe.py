"""k_threshold's two modes (0.115 / 0.127 ms at 2707x181x360) against WHERE and HOW the slab was allocated: one process, the
slab allocated again and again behind dummy allocations of different sizes, with exact and rounded-up sizes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrack_amd import _native
T, ny, nx = 2707, 181, 360
n = T * ny * nx * 4
trk = _native.Tracker(0)
lat = np.linspace(90, -90, ny).astype(np.float32)
w = (111 * 111 * np.cos(lat * np.pi / 180)).astype(np.float32)
thr = np.full(T, 160.0)
trk.set_timing(1)
def measure(d_in, d_out):
    trk.synth_fill(d_in, T, ny, nx, seed=0)
    for _ in range(3):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
    trk.timing_sums(reset=True)
    for _ in range(24):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
    per, _ = trk.timing_sums(reset=True)
    return per["k_threshold"], per["k_relabel"]
out = trk.malloc(n)
for dummy_mb, size in [(0, n), (1, n), (3, n), (37, n), (0, (n + (2 << 20) - 1) & ~((2 << 20) - 1)), (0, (n + (1 << 30) - 1) & ~((1 << 30) - 1)), (129, n), (511, n), (1023, n), (0, n + 4096), (0, n)]:
    d = trk.malloc(dummy_mb << 20) if dummy_mb else None
    a = trk.malloc(size)
    t, r = measure(a, out)
    print("dummy %5d MB  size %11d  va %#x (mod 2MB %#x, mod 1GB %#x)  thr %.4f  rel %.4f" % (dummy_mb, size, a.value, a.value & ((2 << 20) - 1), a.value & ((1 << 30) - 1), t, r))
    trk.free(a)
    if d is not None:
        trk.free(d)
