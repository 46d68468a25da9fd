"""Round 5: counters of k_threshold_v7 on a SLOW handle against a FAST one in the same process (tuner off: CTK_MASK_TUNE=0).
Eight handles are timed (HIP events); then the slowest runs NP passes, then the fastest NP passes: under rocprofv3 --pmc the last
2 * NP dispatches of k_threshold_v7 are [slow] * NP + [fast] * NP (tools/mask_mode_pmc.sh sums them up)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrack_amd import _native
T, ny, nx = [int(x) for x in os.environ.get("SHAPE", "2707,181,360").split(",")]
NP = int(os.environ.get("NP", "6"))
n = T * ny * nx * 4
lat = np.linspace(90, -90, ny).astype(np.float32)
w = (111 * 111 * np.cos(lat * np.pi / 180)).astype(np.float32)
thr = np.full(T, 160.0)
t0 = _native.Tracker(0)
d_in, d_out = t0.malloc(n), t0.malloc(n)
t0.synth_fill(d_in, T, ny, nx, seed=0)
hs, ms = [], []
for k in range(8):
    trk = _native.Tracker(0)
    trk.set_timing(1)
    for _ in range(3):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
    trk.timing_sums(reset=True)
    for _ in range(12):
        trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
    per, _ = trk.timing_sums(reset=True)
    hs.append(trk)
    ms.append(per["k_threshold"])
slow, fast = int(np.argmax(ms)), int(np.argmin(ms))
print("MODES thr ms per handle: %s | slow = %d (%.4f) fast = %d (%.4f)" % (" ".join("%.4f" % x for x in ms), slow, ms[slow], fast, ms[fast]), flush=True)
for k in (slow, fast):
    hs[k].set_timing(0)
    for _ in range(NP):
        hs[k].track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
