"""Timing of the run_lifecycle reductions (ctk_lifecycle_f32_dev) on the bench slab, flags from the GPU tracker;
the scipy port (oracle/lifecycle_port.py) is timed on the first steps for scale.  python tools/lifecycle_probe.py [T ny nx]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrack_amd import _native, synth  # noqa: E402
from contrack_amd.contrack import row_weights, lifecycle_frame, lifecycle_columns  # noqa: E402

T, ny, nx = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2707, 181, 360)
lat, lon = synth.grid(ny, nx)
wrow = row_weights(lat, 180.0 / (ny - 1), 360.0 / nx)
with _native.Tracker(0) as trk:
    n = T * ny * nx
    a, f = trk.malloc(n * 4), trk.malloc(n * 4)
    trk.synth_fill(a, T, ny, nx, 1)
    tracked = trk.track_dev(a, T, ny, nx, np.full(T, 160.0), 0, wrow, 0.5, 5, True, f)
    for _ in range(2):
        rows = trk.lifecycle_dev(f, a, T, ny, nx, wrow)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        rows = trk.lifecycle_dev(f, a, T, ny, nx, wrow)
    dt = (time.perf_counter() - t0) / reps
    dates = ["%06d" % t for t in range(T)]
    t1 = time.perf_counter()
    cols = lifecycle_columns(rows, lat, lon, dates, trk)          # incl. the rows re-evaluated in the reference's summation order
    dt_frame = time.perf_counter() - t1
    frame = lifecycle_frame(rows, lat, lon, dates, trk)
    print("tracked %d contours; lifecycle rows %d (%d rolled); device reductions + download + sort %.3f ms (%.0f timesteps/s); host finish %.1f ms"
          % (tracked, len(rows), int((rows["shift"] > 0).sum()), dt * 1e3, T / dt, dt_frame * 1e3))
    Ts = min(T, 40)
    flag = np.empty((Ts, ny, nx), np.int32)
    anom = np.empty((Ts, ny, nx), np.float32)
    trk.d2h(flag, f)
    trk.d2h(anom, a)
    trk.free(a)
    trk.free(f)
from oracle import lifecycle_port  # noqa: E402
t0 = time.perf_counter()
want = lifecycle_port.run_lifecycle(flag, anom, lat, lon, wrow, ["%06d" % t for t in range(Ts)])
dt_cpu = time.perf_counter() - t0
got = [r for r in frame if int(r[1]) < Ts]
same = len(got) == len(want) and all(a[:4] == b[:4] and abs(a[4] - b[4]) < 0.011 and abs(a[5] - b[5]) < 0.011 for a, b in zip(got, want))
print("scipy port: %d steps in %.2f s (%.1f timesteps/s); rows agree on those steps: %s" % (Ts, dt_cpu, Ts / dt_cpu, same))
