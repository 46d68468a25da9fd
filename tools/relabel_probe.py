"""Paired, in-process A/B of the write kernel's shape: threads and rows per workgroup of k_relabel_v5 (ctk_debug_set_relabel).
python tools/relabel_probe.py [workload] [rounds] [combos "threads:rows,..."]   (0 = the library's default)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from contrack_amd import _native, synth
name = sys.argv[1] if len(sys.argv) > 1 else "era5_025deg_2k"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
combos = [tuple(int(v) for v in c.split(":")) for c in (sys.argv[3] if len(sys.argv) > 3 else "0:0,512:0,1024:0,512:8,1024:8,1024:12").split(",")]
wl = bench.WORKLOADS[name]
T, ny, nx = wl["T"], wl["ny"], wl["nx"]
trk = _native.Tracker(0)
d_in, d_out = trk.malloc(T * ny * nx * 4), trk.malloc(T * ny * nx * 4)
trk.synth_fill(d_in, T, ny, nx, seed=0)
lat, _ = synth.grid(ny, nx)
w = bench.row_weights(lat, np.float32(180.0 / (ny - 1)), np.float32(360.0 / nx))
thr = np.full(T, np.float64(np.float32(160.0)))
L = _native.lib()
trk.set_timing(1)
step = lambda: trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, wl["persistence"], True, d_out)
ref = None
res = {c: [] for c in combos}
for c in combos:                                  # same result whatever the shape
    _native.check(L.ctk_debug_set_relabel(trk.handle, c[0], c[1]))
    n = step(); step()
    cs = trk.checksum_i32(d_out, T * ny * nx)
    ref = ref or (n, cs)
    assert (n, cs) == ref, (c, n, cs, ref)
for r in range(rounds):
    for c in combos:
        _native.check(L.ctk_debug_set_relabel(trk.handle, c[0], c[1]))
        trk.timing_sums(reset=True)
        for _ in range(4):
            step()
        per, cnt = trk.timing_sums(reset=True)
        if cnt["k_relabel"]:
            res[c].append(per["k_relabel"])
base = np.median(res[combos[0]])
px4 = 4.0 * T * ny * nx
print(name, " ".join("%d:%d %.4f ms %.2f TB/s (%+.1f%%)" % (c[0], c[1], np.median(res[c]), px4 / np.median(res[c]) / 1e9, 100 * (np.median(res[c]) / base - 1)) for c in combos))
