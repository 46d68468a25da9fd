"""e2e of the host entry at 2707x181x360 with the result as runs: lanes / blocks per lane (env read once per process)."""
import sys, time, numpy as np
from contrack_amd import _native, synth
T, ny, nx = 2707, 181, 360
if len(sys.argv) > 3: T, ny, nx = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
a = synth.smooth_field(min(T, 300), ny, nx, seed=3)
a = np.ascontiguousarray(np.concatenate([a] * ((T + a.shape[0] - 1) // a.shape[0]))[:T])
w = np.cos(np.deg2rad(np.linspace(-89, 89, ny))).astype(np.float32)
t = _native.Tracker(0)
thr = np.full(T, 160.0)
for mode in (1, 0, 1):
    t.set_result_transfer(mode)
    for k in range(3): f, n = t.track(a, thr, 0, w, 0.5, 5, True); del f
    ts = []
    for k in range(8):
        t0 = time.perf_counter(); f, n = t.track(a, thr, 0, w, 0.5, 5, True); ts.append(time.perf_counter() - t0); tm = t.timings(); del f
    print("mode", mode, "recycled: call %.2f ms" % (1e3 * np.median(ts)), "result leg %.2f ms" % tm["d2h"], "n", n, flush=True)
    keep = []; ts = []
    for k in range(6):
        t0 = time.perf_counter(); f, n = t.track(a, thr, 0, w, 0.5, 5, True); ts.append(time.perf_counter() - t0); tm = t.timings(); keep.append(f)
    print("mode", mode, "fresh:    call %.2f ms" % (1e3 * np.median(ts)), "result leg %.2f ms" % tm["d2h"], flush=True)
    del keep
