"""What the replicated part of the time-sharded path costs when N members of the bench slab are concatenated on the
time axis (weak scaling, BASELINE.json configs[4] style): single GPU, whole concatenated slab; the resolver / seam
driver timers are what EVERY rank would spend at world size N.  python tools/weak_probe.py [N]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrack_amd import _native, synth  # noqa: E402
from contrack_amd.contrack import row_weights  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T, ny, nx = 2707, 181, 360
lat, _ = synth.grid(ny, nx)
w = row_weights(lat, np.float32(1.0), np.float32(1.0))
with _native.Tracker(0) as trk:
    d_in, d_out = trk.malloc(N * T * ny * nx * 4), trk.malloc(N * T * ny * nx * 4)
    import ctypes as C
    for m in range(N):
        a = synth.smooth_field(T, ny, nx, seed=m)
        trk.h2d(C.c_void_p(d_in.value + m * a.nbytes), a)
    thr = np.full(N * T, 160.0)
    trk.set_timing(True)
    for _ in range(3):
        n = trk.track_dev(d_in, N * T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
    acc = {}
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        n = trk.track_dev(d_in, N * T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
        for k, v in trk.timings().items():
            acc[k] = acc.get(k, 0) + v / reps
    dt = (time.perf_counter() - t0) / reps
    print("N=%d members: %d steps, %.3f ms/pass, %.0f timesteps/s, tracked %d" % (N, N * T, dt * 1e3, N * T / dt, n))
    print({k: round(v, 3) for k, v in acc.items()})
    print(trk.stats())
