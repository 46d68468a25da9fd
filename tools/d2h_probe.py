"""Where does the result go fastest?  python tools/d2h_probe.py [T ny nx]
(1) fresh np.empty (first-touch page faults), (2) a reused, already-touched pageable array, (3) ctk_host_alloc'ed pinned memory,
(4) a touched pageable array registered with ctk_host_register -- and what the allocations / registrations themselves cost."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrack_amd import _native, synth
from contrack_amd.contrack import row_weights
T, ny, nx = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2707, 181, 360)
a = synth.smooth_field(T, ny, nx, seed=0)
lat, _ = synth.grid(ny, nx)
w = row_weights(lat, np.float32(180.0 / (ny - 1)), np.float32(360.0 / nx))
thr = np.full(T, 160.0)
L = _native.lib()
nb = T * ny * nx * 4


def run(trk, out, reps=4):
    dt, d2h, h2d = 0.0, 0.0, 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        f, n = trk.track(a, thr, 0, w, 0.5, 5, True, out=out() if callable(out) else out)
        dt += (time.perf_counter() - t0) / reps
        tm = trk.timings()
        d2h += tm["d2h"] / reps; h2d += tm["h2d"] / reps
    return dt * 1e3, h2d, d2h, f, n


with _native.Tracker(0) as trk:
    ref, n0 = trk.track(a, thr, 0, w, 0.5, 5, True)
    trk.track(a, thr, 0, w, 0.5, 5, True)
    keep = []

    def fresh():
        keep.append(np.empty((T, ny, nx), np.int32))
        return keep[-1]
    ms, h2d, d2h, f, n = run(trk, fresh)
    print("fresh np.empty          : %.1f ms per call, H2D %.1f ms, D2H %.1f ms = %.1f GB/s" % (ms, h2d, d2h, nb / d2h / 1e6))
    del keep[:]
    buf = np.empty((T, ny, nx), np.int32); buf[...] = 0
    ms, h2d, d2h, f, n = run(trk, buf)
    print("reused pageable (touched): %.1f ms per call, H2D %.1f ms, D2H %.1f ms = %.1f GB/s   equal %s" % (ms, h2d, d2h, nb / d2h / 1e6, np.array_equal(f, ref)))
    t0 = time.perf_counter()
    p = C.c_void_p()
    _native.check(L.ctk_host_alloc(trk.handle, C.byref(p), nb))
    t_alloc = time.perf_counter() - t0
    pin = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(T, ny, nx))
    ms, h2d, d2h, f, n = run(trk, pin)
    print("ctk_host_alloc (pinned)  : %.1f ms per call, H2D %.1f ms, D2H %.1f ms = %.1f GB/s   equal %s   [allocation %.1f ms]" % (
        ms, h2d, d2h, nb / d2h / 1e6, np.array_equal(f, ref), t_alloc * 1e3))
    t0 = time.perf_counter(); s = int(pin[::7].sum()); print("  CPU read of the pinned result (every 7th step): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    t0 = time.perf_counter(); s = int(buf[::7].sum()); print("  CPU read of the pageable result (same)        : %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    del pin, f
    t0 = time.perf_counter()
    _native.check(L.ctk_host_free(trk.handle, p))
    print("  ctk_host_free %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    t0 = time.perf_counter()
    _native.check(L.ctk_host_register(trk.handle, buf.ctypes.data, nb))
    t_reg = time.perf_counter() - t0
    ms, h2d, d2h, f, n = run(trk, buf)
    print("registered pageable      : %.1f ms per call, H2D %.1f ms, D2H %.1f ms = %.1f GB/s   equal %s   [registration %.1f ms]" % (
        ms, h2d, d2h, nb / d2h / 1e6, np.array_equal(f, ref), t_reg * 1e3))
    t0 = time.perf_counter()
    _native.check(L.ctk_host_unregister(trk.handle, buf.ctypes.data))
    print("  unregister %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    # the input side: a registered input slab
    t0 = time.perf_counter()
    _native.check(L.ctk_host_register(trk.handle, a.ctypes.data, a.nbytes))
    t_reg = time.perf_counter() - t0
    _native.check(L.ctk_host_register(trk.handle, buf.ctypes.data, nb))
    ms, h2d, d2h, f, n = run(trk, buf)
    print("input AND output registered: %.1f ms per call, H2D %.1f ms = %.1f GB/s, D2H %.1f ms = %.1f GB/s  [input registration %.1f ms]" % (
        ms, h2d, a.nbytes / h2d / 1e6, d2h, nb / d2h / 1e6, t_reg * 1e3))
    _native.check(L.ctk_host_unregister(trk.handle, buf.ctypes.data))
    _native.check(L.ctk_host_unregister(trk.handle, a.ctypes.data))
