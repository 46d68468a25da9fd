"""Paired, in-process A/B of the thread counts of the small one-workgroup-per-timestep kernels (ctk_debug_set_small_threads):
python tools/small_probe.py [workload] [rounds]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from contrack_amd import _native, synth
name = sys.argv[1] if len(sys.argv) > 1 else "era5_1deg_djf30"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
wl = bench.WORKLOADS[name]
T, ny, nx = wl["T"], wl["ny"], wl["nx"]
trk = _native.Tracker(0)
d_in, d_out = trk.malloc(T * ny * nx * 4), trk.malloc(T * ny * nx * 4)
if wl.get("device_fill"):
    trk.synth_fill(d_in, T, ny, nx, seed=0)
else:
    trk.h2d(d_in, synth.smooth_field(T, ny, nx, seed=0))
lat, _ = synth.grid(ny, nx)
w = bench.row_weights(lat, np.float32(180.0 / (ny - 1)), np.float32(360.0 / nx))
thr = np.full(T, np.float64(np.float32(160.0)))
L = _native.lib()
step = lambda: trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, wl["persistence"], True, d_out)
combos = [(0, 0, 0), (64, 0, 0), (128, 0, 0), (0, 128, 0), (0, 64, 0), (0, 0, 128), (0, 0, 64), (64, 128, 64)]
ref = None
for c in combos:
    _native.check(L.ctk_debug_set_small_threads(trk.handle, *c))
    n = step(); cs = trk.checksum_i32(d_out, T * ny * nx)
    ref = ref or (n, cs); assert (n, cs) == ref, c
trk.set_timing(2)
res = {c: {"k_extent": [], "k_run_values": [], "k_scan": []} for c in combos}
for r in range(rounds):
    for c in combos:
        _native.check(L.ctk_debug_set_small_threads(trk.handle, *c))
        step()
        tm = trk.timings()
        for k in res[c]:
            res[c][k].append(tm[k])
for c in combos:
    print(name, c, " ".join("%s %.1f us" % (k, 1e3 * np.median(v)) for k, v in res[c].items()))
