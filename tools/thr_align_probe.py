"""Is k_threshold's bimodal time (0.115 / 0.128 ms at 2707x181x360 between process runs) a matter of where the slab lies?
One process, one big allocation, the slab placed at different offsets; HIP-event time of k_threshold per placement."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrack_amd import _native
T, ny, nx = 2707, 181, 360
n = T * ny * nx * 4
trk = _native.Tracker(0)
big = trk.malloc(n + (64 << 20)).value
out = trk.malloc(n).value
lat = np.linspace(90, -90, ny).astype(np.float32)
w = (111 * 111 * np.cos(lat * np.pi / 180)).astype(np.float32)
thr = np.full(T, 160.0)
print("base", hex(big), "out", hex(out))
trk.set_timing(1)
for off in [0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20, 256, 1024, (1 << 20) + 4096, 0, 2 << 20]:
    d_in = big + off
    trk.synth_fill(d_in, T, ny, nx, seed=0)
    for _ in range(3):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, out)
    trk.timing_sums(reset=True)
    for _ in range(24):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, out)
    per, cnt = trk.timing_sums(reset=True)
    print("offset %9d  thr %.4f  rel %.4f" % (off, per["k_threshold"], per["k_relabel"]))
