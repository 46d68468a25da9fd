"""configs[4] (CESM-LE: 40 members x 10 950 daily steps on 192 x 288, concatenated on the time axis = 438 000 steps) at its own size
on one GPU: pass time, stats, one call vs 8 thread-shards.  gpurun: python tools/cesm_probe.py [T]"""
import ctypes as C
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from contrack_amd import _native                      # noqa: E402
from contrack_amd.contrack import row_weights         # noqa: E402
from contrack_amd.dist import shard_bounds            # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 438000
NY, NX = 192, 288
MEMBER = 10950
PLANE = NY * NX


def off(p, nbytes):
    return C.c_void_p(p.value + int(nbytes))


def cesm_lat(ny=NY):
    k = np.arange(ny)
    return (90.0 - 180.0 * (k + 0.5) / ny + 0.3 * np.sin(np.pi * k / (ny - 1))).astype(np.float64)


def main():
    mem = _native.Tracker(0)
    trk = _native.Tracker(0)
    d_in = mem.malloc(T * PLANE * 4)
    d_out = mem.malloc(T * PLANE * 4)
    lat = cesm_lat()
    dlat = np.float64(round(float(np.abs(np.diff(lat)).mean()), 2))
    w = row_weights(lat, dlat, np.float64(360.0 / NX))
    t0 = time.perf_counter()
    for m, tb in enumerate(range(0, T, MEMBER)):
        n = min(MEMBER, T - tb)
        trk.synth_fill(off(d_in, tb * PLANE * 4), n, NY, NX, seed=100 + m)
    print("fill %.2f s" % (time.perf_counter() - t0), flush=True)
    thr = np.full(T, np.float64(np.float32(160.0)))
    trk.memset(d_out, 0xff, T * PLANE * 4)
    for rep in range(4):
        t0 = time.perf_counter()
        n_one = trk.track_dev(d_in, T, NY, NX, thr, 0, w, 0.5, 5, True, d_out)
        dt = time.perf_counter() - t0
        st = trk.stats()
        print("pass %d: %.2f ms  n_tracked %d  %.3f of 8 TB/s  fused %d reason %d" % (rep, dt * 1e3, n_one, 8.0 * T * PLANE / dt / 8e12, st["fused_pass"], st["off_fused_path_reason"]), flush=True)
    print(st, flush=True)
    trk.set_timing(2)
    trk.track_dev(d_in, T, NY, NX, thr, 0, w, 0.5, 5, True, d_out)
    print({k: round(v, 3) for k, v in trk.timings().items() if v}, flush=True)
    trk.set_timing(0)
    pr = trk.check_flag(d_in, d_out, T, NY, NX, thr, 0, 5, st["labels_3d"])
    print(pr, flush=True)
    world = 8
    bounds = shard_bounds(T, world)
    ref = [trk.checksum_i32(off(d_out, a * PLANE * 4), (b - a) * PLANE, a * PLANE) for a, b in bounds]
    trk.memset(d_out, 0xff, T * PLANE * 4)
    trk.close()
    trk = mem
    hs = [_native.Tracker(0) for _ in range(world)]
    group = _native.CommGroup(world)
    comms = [_native.Comm.local(hs[r], group, r) for r in range(world)]
    res, err = [None] * world, [None] * world

    def work(r):
        a, b = bounds[r]
        try:
            res[r] = hs[r].track_sharded_dev(comms[r], off(d_in, a * PLANE * 4), b - a, a, T, NY, NX, thr[a:b].copy(), 0, w, 0.5, 5, True, off(d_out, a * PLANE * 4))
        except Exception as e:      # noqa: BLE001
            err[r] = e

    for rep in range(2):
        th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        print("8 thread-shards on one GPU: %.2f ms" % ((time.perf_counter() - t0) * 1e3), res, err, flush=True)
    got = [trk.checksum_i32(off(d_out, a * PLANE * 4), (b - a) * PLANE, a * PLANE) for a, b in bounds]
    print("checksums equal:", got == ref, " n equal:", res == [n_one] * world)
    print([h.stats()["off_fused_path_reason"] for h in hs], [h.stats()["filter_rounds"] for h in hs])


main()
