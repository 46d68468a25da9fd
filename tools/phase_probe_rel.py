"""Stamps of one workgroup in the middle of k_relabel_v5's grid on a bench workload (library built with -DCTK_PHASE_TIMING into tools/exp/lib_phase.so):
entry | guard read | tables in LDS | first image: zeroed | decoded | stored (issued) | barrier behind it | end.   python tools/phase_probe_rel.py [workload]"""
import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CTK_LIB"] = os.path.join(ROOT, "tools", "exp", "lib_phase.so")
import bench
from contrack_amd import _native
name = sys.argv[1] if len(sys.argv) > 1 else "era5_1deg_djf30"
wl = bench.WORKLOADS[name]
T, ny, nx = wl["T"], wl["ny"], wl["nx"]
trk = _native.Tracker(0)
nbytes = T * ny * nx * 4
d_in, d_out = trk.malloc(nbytes), trk.malloc(nbytes)
w = bench.workload_weights(wl)
if wl.get("device_fill"):
    bench.device_fill(trk, d_in, wl)
else:
    a, _ = bench.make_slab(wl)
    trk.h2d(d_in, a)
thr = np.full(T, np.float64(np.float32(wl["threshold"])))
op = _native.CMP_OPS[wl["gorl"]]
L = _native.lib()
buf = (C.c_ulonglong * 16)()
acc = (C.c_ulonglong * 2048)()
L.ctk_debug_rel_acc.argtypes = [C.c_void_p, C.c_int]
for rep in range(8):
    L.ctk_debug_rel_acc(None, 1)
    trk.set_timing(2)
    trk.track_dev(d_in, T, ny, nx, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"], d_out)
    if rep < 4:
        continue
    L.ctk_debug_rel_acc(acc, 0)
    v = np.array(list(acc), dtype=np.float64)
    kms = trk.timings()["k_relabel"]
    rb = trk.stats()
    print("all workgroups: sum of (end - entry) %.1f us, longest %.2f us; kernel %.1f us -> %.1f workgroups in their code at a time, %.2f per CU" % (
        v[:1024].sum() / 100.0, v[1024:].max() / 100.0, kms * 1e3, v[:1024].sum() / 100.0 / (kms * 1e3), v[:1024].sum() / 100.0 / (kms * 1e3) / 256))
    L.ctk_debug_rel_times(buf)
    t = np.array(list(buf), dtype=np.int64)
    names = ["guard read", "tables in LDS (one trip + barrier)", "(set-up)", "image zeroed + barrier", "decoded + barrier", "stores issued", "barrier behind the stores", "second image ... end"]
    print("us: " + " | ".join("%s %.2f" % (names[i], (t[i + 1] - t[i]) / 100.0) for i in range(8)) + " | whole workgroup %.2f" % ((t[8] - t[0]) / 100.0))
