"""Paired A/B of the chunk -> XCD mapping of the two streaming kernels inside ONE process (the board drifts between processes and
over seconds: only interleaved passes compare).  python tools/xcd_probe.py [workload] [rounds]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from contrack_amd import _native, synth
name = sys.argv[1] if len(sys.argv) > 1 else "era5_1deg_djf30"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
modes = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 4, 16, 64, 256]
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
for _ in range(8):
    step()
res = {("thr", m): [] for m in modes}
res.update({("rel", m): [] for m in modes})
for which in ("thr", "rel"):
    for r in range(rounds):
        for m in modes:
            L.ctk_debug_set_xcd(trk.handle, m if which == "thr" else 0, m if which == "rel" else 0)
            trk.timing_sums(reset=True)
            for _ in range(4):                      # level-1 timing: k_threshold in passes 0 mod 4, k_relabel in passes 2 mod 4
                step()
            per, cnt = trk.timing_sums(reset=True)
            k = "k_threshold" if which == "thr" else "k_relabel"
            if cnt[k]:
                res[(which, m)].append(per[k])
for which in ("thr", "rel"):
    base = np.median(res[(which, modes[0])])
    print(name, which, " ".join("%d: %.4f (%+.1f%%)" % (m, np.median(res[(which, m)]), 100 * (np.median(res[(which, m)]) / base - 1)) for m in modes))
