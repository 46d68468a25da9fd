"""tools/shard_probe.py -- what the time-sharded product path costs, measured on ONE GPU.
  (a) ctk_track_sharded_* with a world of one vs the fused one-call path on the bench slab (the orchestration overhead: extra
      host hand-offs, boundary kernels);
  (b) N members concatenated on the time axis, N ranks as threads sharing the GPU: how much of the seam work is shared
      (driven on every rank) and how much stays local; bit-exactness against the one-call result."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from contrack_amd import _native, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T, ny, nx = 2707, 181, 360
lat, _ = synth.grid(ny, nx)
w = np.array((111 * np.float32(1.0) * 111 * np.float32(1.0) * np.cos(lat * np.pi / 180))).astype(np.float32)
thr1 = np.full(T, np.float64(np.float32(160.0)))

# (a)
trk = _native.Tracker(0)
a0 = synth.smooth_field(T, ny, nx, seed=0)
d_in, d_out = trk.malloc(a0.nbytes), trk.malloc(a0.nbytes)
trk.h2d(d_in, a0)
g1 = _native.CommGroup(1)
c1 = _native.Comm.local(trk, g1, 0)
for name, fn in (("fused", lambda: trk.track_dev(d_in, T, ny, nx, thr1, 0, w, 0.5, 5, True, d_out)),
                 ("sharded, world 1", lambda: trk.track_sharded_dev(c1, d_in, T, 0, T, ny, nx, thr1, 0, w, 0.5, 5, True, d_out))):
    for _ in range(3):
        n = fn()
    trk.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        n = fn()
    trk.sync()
    print("%-18s %.3f ms / pass, n_tracked %d" % (name, (time.perf_counter() - t0) * 50, n))
c1.close(); g1.close()
trk.free(d_in); trk.free(d_out)

# (b)
members = [synth.smooth_field(T, ny, nx, seed=s) for s in range(N)]
trks = [_native.Tracker(0) for _ in range(N)]
group = _native.CommGroup(N)
comms = [_native.Comm.local(trks[r], group, r) for r in range(N)]
bufs = []
for r in range(N):
    di, do = trks[r].malloc(members[r].nbytes), trks[r].malloc(members[r].nbytes)
    trks[r].h2d(di, members[r])
    bufs.append((di, do))
res, stats, times = [None] * N, [None] * N, [None] * N


def work(r, reps):
    for _ in range(reps):
        t0 = time.perf_counter()
        res[r] = trks[r].track_sharded_dev(comms[r], bufs[r][0], T, r * T, N * T, ny, nx, thr1, 0, w, 0.5, 5, True, bufs[r][1])
        times[r] = time.perf_counter() - t0
    stats[r] = trks[r].stats()


for reps in (2, 5):
    th = [threading.Thread(target=work, args=(r, reps)) for r in range(N)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
print("%d members x %d steps, %d ranks sharing one GPU: %.2f ms per pass (all ranks, one GPU), n_tracked %s" % (N, T, N, wall * 1e3 / 5, set(res)))
for r in range(N):
    s = stats[r]
    print("rank %d: seam rows %5d of which shared %4d | ops %5d | labels(all) %6d | filter rounds %d passes %d | host seam loops %.3f ms" % (
        r, s["seam_rows_to_driver"], s["shared_seam_rows"], s["seam_ops"], s["labels_3d"], s["filter_rounds"], s["filter_passes"], s["seam_loop_ns"] / 1e6))
# bit-exactness against ONE call on the concatenated slab
big = np.concatenate(members, axis=0)
ref = _native.Tracker(0)
want, nw = ref.track(big, np.full(N * T, np.float64(np.float32(160.0))), 0, w, 0.5, 5, True)
ok = True
for r in range(N):
    f = np.empty((T, ny, nx), np.int32)
    trks[r].d2h(f, bufs[r][1])
    ok = ok and np.array_equal(f, want[r * T:(r + 1) * T])
print("bit-exact vs one call on the concatenated slab:", ok, "| n_tracked", nw)
