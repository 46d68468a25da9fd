"""configs[4] grid (192 x 288), T steps (default 438 000): time of the write kernel against rows / threads per workgroup
(ctk_debug_set_relabel), and of every kernel group at the default.  gpurun: python tools/cesm_relabel_sweep.py [T]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import bench                                           # noqa: E402
from contrack_amd import _native                      # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 438000
wl = dict(bench.WORKLOADS[os.environ.get("WL", "cesm_le_40x30yr")])
if len(sys.argv) > 1:
    wl = dict(wl, T=T, members=max(1, T // 10950))
T = wl["T"]
ny, nx = wl["ny"], wl["nx"]
trk = _native.Tracker(0)
d_in, d_out = trk.malloc(T * ny * nx * 4), trk.malloc(T * ny * nx * 4)
bench.device_fill(trk, d_in, wl)
w = bench.workload_weights(wl)
thr = np.full(T, np.float64(np.float32(160.0)))


def run(reps=3):
    acc = {}
    for _ in range(reps):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
        for k, v in trk.timings().items():
            acc[k] = acc.get(k, 0.0) + v / reps
    return acc


trk.set_timing(2)
run(2)
base = run()
print("default:", {k: round(v, 3) for k, v in base.items() if v}, flush=True)
for threads in (0,):
    for rows in [int(x) for x in os.environ.get("ROWS", "48,64,96").split(",")]:
        _native.check(_native.lib().ctk_debug_set_relabel(trk.handle, threads, rows))
        try:
            r = run(2)
            print("threads %4d rows %3d: k_relabel %.3f ms  (%.2f TB/s)  kernel %s" % (threads or 256, rows, r["k_relabel"], 4.0 * T * ny * nx / r["k_relabel"] / 1e9, trk.stats()["relabel_kernel"]), flush=True)
        except Exception as e:      # noqa: BLE001
            print("threads %d rows %d: %s" % (threads, rows, e), flush=True)
_native.check(_native.lib().ctk_debug_set_relabel(trk.handle, 0, 0))
