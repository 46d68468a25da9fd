"""Phase times of k_life_exact's workgroup with the largest contour (> 50 000 pixels; library built with -DCTK_PHASE_TIMING into
tools/exp/lib_phase.so): pairwise sums and, next to them, the sequential sums.  python tools/phase_probe_life.py 480 721 1440"""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CTK_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "lib_phase.so")
from contrack_amd import _native, synth
from contrack_amd.contrack import row_weights, lifecycle_columns
T, ny, nx = (int(v) for v in sys.argv[1:4])
lat, lon = synth.grid(ny, nx)
wrow = row_weights(lat, 180.0 / (ny - 1), 360.0 / nx)
trk = _native.Tracker(0)
n = T * ny * nx
a, f = trk.malloc(n * 4), trk.malloc(n * 4)
trk.synth_fill(a, T, ny, nx, 1)
trk.track_dev(a, T, ny, nx, np.full(T, 160.0), 0, wrow, 0.5, 5, True, f)
for rep in range(3):
    rows = trk.lifecycle_dev(f, a, T, ny, nx, wrow)
    lifecycle_columns(rows, lat, lon, ["%06d" % t for t in range(T)], trk)
    buf = (C.c_ulonglong * 16)()
    _native.lib().ctk_debug_phase_times(buf)
    t = np.array(list(buf), dtype=np.int64)
    print("us from the entry of k_life_exact's workgroup: pairwise sums done after %.1f, sequential sums after %.1f, both (workgroup) after %.1f" % (
        (t[2] - t[1]) / 100.0, (t[4] - t[1]) / 100.0, (t[3] - t[1]) / 100.0))
