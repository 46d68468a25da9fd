#!/usr/bin/env python3
"""Round 6: what would pipelining the FRONT of the pass over two sub-slabs buy (verdict item 2)?

The front (threshold -> row counts -> run scan -> 2-D labelling -> co-occurrence) of the two halves of the bench slab on two handles
(own streams, own work spaces) driven by two host threads -- against the two halves one after the other on one handle, and against
the whole slab in one go.  The staged entries (ctk_shard_label2d + ctk_shard_overlap) are used: same kernels as the one-call pass up
to k_overlap (compaction by scan + k_compact_comps).  Device-generated slab (k_synth); ms per iteration over 30 iterations.
"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from contrack_amd import _native

T, ny, nx = 2707, 181, 360
w = bench.workload_weights(dict(ny=ny, nx=nx))
thr = np.full(T, np.float64(np.float32(160.0)))
base = _native.Tracker(0)
d_in = base.malloc(T * ny * nx * 4)
base.synth_fill(d_in, T, ny, nx, seed=0)
base.sync()
import ctypes as C
Th = (T + 1) // 2
halves = [(0, Th), (Th, T - Th)]


def front(trk, t0, nt):
    trk.shard_label2d(C.c_void_p(d_in.value + t0 * ny * nx * 4), nt, ny, nx, thr[t0:t0 + nt], 0, w, 0)
    trk.shard_overlap()


def timed(fn, n=30):
    for _ in range(5):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) * 1e3 / n


one = _native.Tracker(0)
def whole():
    front(one, 0, T); one.sync()
print("whole slab, one handle:                    %.4f ms" % timed(whole))
def serial():
    for t0, nt in halves:
        front(one, t0, nt)
    one.sync()
print("two halves one after the other, one handle: %.4f ms" % timed(serial))
def half_only():
    front(one, 0, Th); one.sync()
print("one half alone:                             %.4f ms" % timed(half_only))
for nh in (2, 4):
    trks = [_native.Tracker(0) for _ in range(nh)]
    per = (T + nh - 1) // nh
    parts = [(k * per, min(per, T - k * per)) for k in range(nh)]
    for (t0, nt), t in zip(parts, trks):
        for _ in range(3):
            front(t, t0, nt); t.sync()
    n = 30
    start = threading.Barrier(nh + 1)
    def run(i):
        t0_, nt = parts[i]
        start.wait()
        for _ in range(n):
            front(trks[i], t0_, nt)
            trks[i].sync()
    th = [threading.Thread(target=run, args=(i,)) for i in range(nh)]
    for x in th: x.start()
    start.wait()
    t0 = time.perf_counter()
    for x in th: x.join()
    dt = (time.perf_counter() - t0) * 1e3 / n
    print("%d parts on %d handles / host threads at once:  %.4f ms per round (free-running: every thread syncs its own handle)" % (nh, nh, dt))
    # staggered by construction: part k starts when part k-1 has launched its threshold kernel is not controllable from here
    for t in trks: t.close()
