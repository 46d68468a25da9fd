"""configs[4] (438 000 x 192 x 288): threads per workgroup of the one-workgroup-per-timestep kernels k_extent / k_run_values / k_compact_init
(ctk_debug_set_small_threads) in the throughput regime -- at 2707 steps (latency regime) 256 won for the last two.  gpurun: python tools/cesm_small_sweep.py"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import bench                                           # noqa: E402
from contrack_amd import _native                      # noqa: E402

wl = dict(bench.WORKLOADS[os.environ.get("WL", "cesm_le_40x30yr")])
T, ny, nx = wl["T"], wl["ny"], wl["nx"]
trk = _native.Tracker(0)
d_in, d_out = trk.malloc(T * ny * nx * 4), trk.malloc(T * ny * nx * 4)
bench.device_fill(trk, d_in, wl)
w = bench.workload_weights(wl)
thr = np.full(T, np.float64(np.float32(160.0)))


def run(reps=3):
    acc = {}
    for _ in range(reps):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, wl["overlap"], wl["persistence"], True, d_out)
        for k, v in trk.timings().items():
            acc[k] = acc.get(k, 0.0) + v / reps
    return acc


trk.set_timing(2)
run(3)
for cfg in ((0, 0, 0), (64, 64, 64), (128, 128, 128), (64, 128, 128), (64, 256, 128), (64, 128, 256)):
    _native.check(_native.lib().ctk_debug_set_small_threads(trk.handle, *cfg))
    run(1)
    r = run(3)
    print("extent / run_values / compact_init threads %s: k_extent %.3f  k_run_values %.3f  k_scan (rowcount + scan + compact_init) %.3f  total %.2f ms" % (
        cfg, r["k_extent"], r["k_run_values"], r["k_scan"], r["total"]), flush=True)
