"""Is k_threshold's mode (0.106 or 0.119 ms at 2707x181x360, per handle) a matter of where the handle's bit mask lies?  One handle, the
mask moved through offsets inside one allocation (ctk_debug_set_mask_offset), interleaved rounds; then several handles at offset -1."""
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
offs = [-1, 0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 12 << 20, 16 << 20, 24 << 20, 32 << 20, 48 << 20]
res = {o: [] for o in offs}
for rnd in range(3):
    for o in offs:
        _native.check(L.ctk_debug_set_mask_offset(trk.handle, o))
        for _ in range(3):
            trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
        trk.timing_sums(reset=True)
        for _ in range(16):
            trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
        per, _ = trk.timing_sums(reset=True)
        res[o].append((per["k_threshold"], per["k_relabel"]))
for o in offs:
    print("mask offset %9d: thr %s   rel %s" % (o, " ".join("%.4f" % a for a, _ in res[o]), " ".join("%.4f" % b for _, b in res[o])))
