"""The write kernel's variants timed in ONE process on the finished tables of a bench workload, launch by launch in turn
(ctk_debug_time_relabel): k_relabel_v5 | the same without its SGPR limit, each in the chunk -> XCD orders 0 / 1 / 16.
    python tools/relabel_variants.py [workload] [rounds]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from contrack_amd import _native
name = sys.argv[1] if len(sys.argv) > 1 else "era5_1deg_djf30"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl = bench.WORKLOADS[name]
T, ny, nx = wl["T"], wl["ny"], wl["nx"]
trk = _native.Tracker(0)
nbytes = T * ny * nx * 4
d_in, d_out = trk.malloc(nbytes), trk.malloc(nbytes)
w = bench.workload_weights(wl)
if wl.get("device_fill"):
    bench.device_fill(trk, d_in, wl)
else:
    a, _ = bench.make_slab(wl)
    trk.h2d(d_in, a)
thr = np.full(T, np.float64(np.float32(wl["threshold"])))
op = _native.CMP_OPS[wl["gorl"]]
for _ in range(3):
    n = trk.track_dev(d_in, T, ny, nx, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"], d_out)
ref = trk.checksum_i32(d_out, T * ny * nx)
names = {0: "k_relabel_v5 (80 SGPRs)", 1: "k_relabel_v5, all SGPRs"}
res = {}
for r in range(rounds):
    for xcd in (0, 1, 16):
        for v in (0, 1):
            ms = trk.time_relabel(d_out, wl["persistence"], v, xcd, reps=3)
            res.setdefault((v, xcd), []).extend(ms[1:].tolist())
            if r == 0:
                cs = trk.checksum_i32(d_out, T * ny * nx)
                assert tuple(cs) == tuple(ref), ("checksum", v, xcd)
print("%s: %d x %d x %d, %d launches per cell (us: min / median); checksums equal" % (name, T, ny, nx, 2 * rounds))
for v in (0, 1):
    print("  %-26s " % names[v] + " | ".join("xcd %2d: %6.1f / %6.1f" % (x, 1e3 * min(res[(v, x)]), 1e3 * float(np.median(res[(v, x)]))) for x in (0, 1, 16)))
