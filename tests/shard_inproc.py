"""The staged (time-sharded) C API driven in ONE process: N handles on one GPU play N ranks; the exchanges are plain
device pointers and numpy reductions (no torch, no processes).  Used by tests/test_gpu_sharded_inprocess.py and
tools/fuzz_sharded.py."""
import numpy as np

from contrack_amd import _native


def sharded(trks, a, thrv, op, w, ov, pers, two, cuts, device_resolve):
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
    exts = []
    if device_resolve:
        blobs = [trks[r].shard_tables_dev() for r in range(n)]
        for r in range(n):
            trks[r].sync()
        for r in range(n):
            ext, nl = trks[r].shard_resolve_dev([b[0].value for b in blobs], [b[1] for b in blobs], r, cuts[r], ov, two)
            trks[r].sync()
            exts.append((ext, nl))
    else:
        host = [trks[r].shard_tables() for r in range(n)]
        res = _native.resolve(host, ov, two)
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
    return np.concatenate(out, axis=0), alive + (1 if bg else 0) - 1
