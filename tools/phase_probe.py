import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CTK_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "lib_phase.so")
from contrack_amd import _native, synth
T, ny, nx = (int(v) for v in sys.argv[1:4])
a = synth.smooth_field(T, ny, nx, seed=0)
lat, _ = synth.grid(ny, nx)
w = np.array(111 * np.float32(180 / (ny - 1)) * 111 * np.float32(360 / nx) * np.cos(lat * np.pi / 180)).astype(np.float32)
trk = _native.Tracker(0)
thr = np.full(T, np.float64(np.float32(160)))
for _ in range(3):
    trk.track(a, thr, 0, w, 0.5, 5, True)
buf = (C.c_ulonglong * 16)()
_native.lib().ctk_debug_phase_times(buf)
t = np.array(list(buf), dtype=np.int64)
print("phase durations (us, wall_clock64 @100MHz):", [(i, (t[i + 1] - t[i]) / 100.0) for i in range(8)])
print("entry of the probed workgroup -> phase mark 0: %.2f us; first workgroup's entry -> probed one's: %.2f us; -> last one's: %.2f us; probed workgroup entry -> its last mark: %.2f us" % (
    (t[0] - t[15]) / 100.0, (t[15] - t[14]) / 100.0, (t[13] - t[14]) / 100.0, (t[8] - t[15]) / 100.0))
