"""End-to-end time of the host-array entry (numpy in, numpy out: what the drop-in class pays), vs the device-resident
call.  python tools/e2e_probe.py [T ny nx]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrack_amd import _native, synth
from contrack_amd.contrack import row_weights
T, ny, nx = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2707, 181, 360)
a = synth.smooth_field(T, ny, nx, seed=0)
lat, _ = synth.grid(ny, nx)
w = row_weights(lat, np.float32(180.0 / (ny - 1)), np.float32(360.0 / nx))
thr = np.full(T, 160.0)
with _native.Tracker(0) as trk:
    for _ in range(2):
        f, n = trk.track(a, thr, 0, w, 0.5, 5, True)
    reps, keep, dt, lib = 5, [], 0.0, {}
    for _ in range(reps):
        t0 = time.perf_counter()
        f, n = trk.track(a, thr, 0, w, 0.5, 5, True)
        dt += time.perf_counter() - t0
        keep.append(f)                    # results stay alive: freeing 700 MB of pages is Python's business, not the call's
        for k, v in trk.timings().items():
            lib[k] = lib.get(k, 0.0) + v / reps
    dt /= reps
    print("host arrays in/out: %.1f ms per call (%.0f timesteps/s; inside the library: H2D %.1f ms = %.1f GB/s, device pass %.2f ms, "
          "D2H %.1f ms = %.1f GB/s; %d tracked)" % (dt * 1e3, T / dt, lib["h2d"], a.nbytes / lib["h2d"] / 1e6, lib["total"] - lib["h2d"] - lib["d2h"],
                                                    lib["d2h"], a.nbytes / lib["d2h"] / 1e6, n))
