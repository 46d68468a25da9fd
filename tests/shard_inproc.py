"""Time shards driven in ONE process, N handles on one GPU playing N ranks:
  sharded          the staged C API (table-level protocol, host resolver); exchanges are plain device pointers / numpy
  sharded_threads  the product path ctk_track_sharded_*: one host thread per rank, in-process communicator group
Used by tests/test_gpu_sharded*.py and tools/fuzz_sharded.py."""
import numpy as np

from contrack_amd import _native


def sharded(trks, a, thrv, op, w, ov, pers, two, cuts):
    """the staged C API with the host resolver; returns (flag, n_tracked, resolver info)"""
    T, ny, nx = a.shape
    n = len(cuts) - 1
    bufs = []
    for r in range(n):
        t0, t1 = cuts[r], cuts[r + 1]
        d_in, d_out = trks[r].malloc(max((t1 - t0) * ny * nx * 4, 8)), trks[r].malloc(max((t1 - t0) * ny * nx * 4, 8))
        trks[r].h2d(d_in, np.ascontiguousarray(a[t0:t1]))
        bufs.append((d_in, d_out))
        trks[r].shard_label2d(d_in, t1 - t0, ny, nx, thrv[t0:t1], op, w, r > 0)
    for r in range(n - 1):
        p, sz = trks[r].halo_export()
        trks[r].sync()
        trks[r + 1].halo_import(p, sz)
    for r in range(n):
        trks[r].shard_overlap()
    host = [trks[r].shard_tables() for r in range(n)]
    res = _native.resolve(host, ov, two)
    info = res.info()
    exts = []
    for r in range(n):
        exts.append(trks[r].shard_extents(res, r, cuts[r]))
        trks[r].sync()
    nl = exts[0][1]
    arrs = []
    for r in range(n):
        e = np.empty(2 * (nl + 1), np.int32)
        trks[r].d2h(e, exts[r][0])
        arrs.append(e)
    tmin = np.min([e[:nl + 1] for e in arrs], axis=0); tmax = np.max([e[nl + 1:] for e in arrs], axis=0)
    red = np.concatenate([tmin, tmax]).astype(np.int32)
    out, alive, bg = [], None, False
    for r in range(n):
        trks[r].h2d(exts[r][0], red)
        na, z = trks[r].shard_write(pers, bufs[r][1])
        f = np.empty((cuts[r + 1] - cuts[r], ny, nx), np.int32)
        trks[r].d2h(f, bufs[r][1])
        out.append(f); alive = na; bg = bg or z
        trks[r].free(bufs[r][0]); trks[r].free(bufs[r][1])
    return np.concatenate(out, axis=0), alive + (1 if bg else 0) - 1, info


def sharded_threads(trks, a, thrv, op, w, ov, pers, two, cuts, f64=False):
    """ctk_track_sharded_*: N handles on one GPU, one host thread per rank, in-process communicator group.  Returns the
    concatenated flag, n_tracked (identical on all ranks, checked) and the per-rank stats."""
    import threading
    T, ny, nx = a.shape
    n = len(cuts) - 1
    group = _native.CommGroup(n)
    comms = [_native.Comm.local(trks[r], group, r) for r in range(n)]
    out, res, err, stats = [None] * n, [None] * n, [None] * n, [None] * n

    def work(r):
        t0, t1 = cuts[r], cuts[r + 1]
        nb = max((t1 - t0) * ny * nx * (8 if f64 else 4), 8)
        d_in, d_out = trks[r].malloc(nb), trks[r].malloc(max((t1 - t0) * ny * nx * 4, 8))
        try:
            trks[r].h2d(d_in, np.ascontiguousarray(a[t0:t1], dtype=np.float64 if f64 else np.float32))
            res[r] = trks[r].track_sharded_dev(comms[r], d_in, t1 - t0, t0, T, ny, nx, thrv[t0:t1], op, w, ov, pers, two, d_out, f64=f64)
            f = np.empty((t1 - t0, ny, nx), np.int32)
            trks[r].d2h(f, d_out)
            out[r] = f
            stats[r] = trks[r].stats()
        except Exception as e:                     # noqa: BLE001 -- reported by the caller
            err[r] = e
        finally:
            trks[r].free(d_in); trks[r].free(d_out)

    th = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c in comms:
        c.close()
    group.close()
    for e in err:
        if e is not None:
            raise e
    assert len(set(res)) == 1, res
    return np.concatenate(out, axis=0), res[0], stats
