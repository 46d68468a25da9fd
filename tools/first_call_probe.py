import sys, numpy as np
sys.path.insert(0, ".")
import bench
from contrack_amd import _native
wl = bench.WORKLOADS[sys.argv[1]]
T, ny, nx = wl["T"], wl["ny"], wl["nx"]
trk = _native.Tracker(0)
d_in, d_out = trk.malloc(T*ny*nx*4), trk.malloc(T*ny*nx*4)
bench.device_fill(trk, d_in, wl)
w = bench.workload_weights(wl)
thr = np.full(T, np.float64(np.float32(160.0)))
if len(sys.argv) > 2: trk.set_timing(int(sys.argv[2]))
for i in range(3):
    n = trk.track_dev(d_in, T, ny, nx, thr, 0, w, wl["overlap"], wl["persistence"], True, d_out)
    st = trk.stats()
    print(i, n, {k: st[k] for k in ("fused_pass","off_fused_path_reason","host_path","filter_passes","pairs","labels_3d","seam_ops","pair_table_regrows")})
