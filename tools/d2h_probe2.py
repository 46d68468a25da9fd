"""fresh result arrays: register-then-DMA against the bounce path.  python tools/d2h_probe2.py"""
import ctypes as C, os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrack_amd import _native, synth
from contrack_amd.contrack import row_weights
T, ny, nx = 2707, 181, 360
a = synth.smooth_field(T, ny, nx, seed=0)
lat, _ = synth.grid(ny, nx)
w = row_weights(lat, np.float32(1.0), np.float32(1.0))
thr = np.full(T, 160.0)
L = _native.lib()
libc = C.CDLL(None, use_errno=True)
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
nb = T * ny * nx * 4
MADV_HUGEPAGE, MADV_POPULATE_WRITE = 14, 23


def huge(arr):
    p = arr.ctypes.data
    a0 = (p + (2 << 20) - 1) & ~((2 << 20) - 1); a1 = (p + arr.nbytes) & ~((2 << 20) - 1)
    if a1 > a0:
        libc.madvise(a0, a1 - a0, MADV_HUGEPAGE)


def populate(arr, nthreads=8):
    p, n = arr.ctypes.data, arr.nbytes
    p0 = p & ~4095
    n = n + (p - p0)
    step = ((n // nthreads) + 4095) & ~4095
    th = [threading.Thread(target=lambda i=i: libc.madvise(p0 + i * step, max(0, min(step, n - i * step)), MADV_POPULATE_WRITE)) for i in range(nthreads)]
    for t in th: t.start()
    for t in th: t.join()


with _native.Tracker(0) as trk:
    ref, n0 = trk.track(a, thr, 0, w, 0.5, 5, True)
    trk.track(a, thr, 0, w, 0.5, 5, True)
    for name, mode in (("bounce (today)", 0), ("register fresh", 1), ("hugepage + register fresh", 2), ("populate x8 + register", 3), ("hugepage + populate x8 + register", 4)):
        keep, tot, treg, td2h = [], 0.0, 0.0, 0.0
        for _ in range(4):
            t0 = time.perf_counter()
            out = np.empty((T, ny, nx), np.int32)
            if mode in (2, 4): huge(out)
            if mode in (3, 4): populate(out)
            t1 = time.perf_counter()
            if mode: _native.check(L.ctk_host_register(trk.handle, out.ctypes.data, nb))
            t2 = time.perf_counter()
            f, n = trk.track(a, thr, 0, w, 0.5, 5, True, out=out)
            td2h += trk.timings()["d2h"] / 4
            if mode: _native.check(L.ctk_host_unregister(trk.handle, out.ctypes.data))
            tot += (time.perf_counter() - t0) / 4; treg += (t2 - t0) / 4
            keep.append(out)
        print("%-36s: %.1f ms per call in all (alloc/populate/register %.1f ms, D2H %.1f ms)  equal %s" % (name, tot * 1e3, treg * 1e3, td2h, np.array_equal(keep[-1], ref)))
        del keep
