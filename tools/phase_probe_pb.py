"""Per-workgroup stamps of k_rs_pass_blk on a bench workload (library built with -DCTK_PHASE_TIMING into tools/exp/lib_phase.so):
entry of the first wave | start-up barrier passed | iterations done (last wave) | end (last wave).
    python tools/phase_probe_pb.py [workload]"""
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
L.ctk_debug_pb_times.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 4096)()
for _ in range(5):
    trk.track_dev(d_in, T, ny, nx, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"], d_out)
for rep in range(3):
    L.ctk_debug_pb_times(None, 1)
    trk.track_dev(d_in, T, ny, nx, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"], d_out)
    L.ctk_debug_pb_times(buf, 0)
    t = np.array(list(buf)[:4096], dtype=np.uint64).reshape(1024, 4)
    nwg = (T + 15) // 16
    t = t[:nwg].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    ent, su, itr, end = us[:, 0], us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
    q = lambda x: "min %.1f med %.1f p90 %.1f max %.1f (wg %d)" % (x.min(), np.median(x), np.percentile(x, 90), x.max(), int(x.argmax()))
    print("rep %d: %d workgroups; first entry -> last end %.1f us" % (rep, nwg, us[:, 3].max()))
    print("   entry after the first one: " + q(ent))
    print("   start-up (loads, barrier): " + q(su))
    print("   iterations:                " + q(itr))
    print("   wait for t-1 + unions:     " + q(end))
    print("   end after the first entry: " + q(us[:, 3]))
    if rep == 2:
        o = np.argsort(-us[:, 3])[:8]
        for i in o:
            print("     wg %4d: entry %.1f  start-up %.1f  iterations %.1f  unions %.1f  end %.1f" % (i, ent[i], su[i], itr[i], end[i], us[i, 3]))
