"""GPU-free parts of the product's time-sharded path (ctk_track_sharded_*, contrack_amd/csrc/ctk_sharded.hip):

* the label numbering across shard boundaries (boundary_resolve in csrc/ctk_seam.h, through the ctk_debug_boundary_resolve
  hook): every rank labels its own shard and ranks its own roots; the boundary records of all ranks must turn that into
  scipy's global raster-order ids (contrack.py:748-751) -- checked against scipy.ndimage.label on random masks, random cuts;
* the launcher-side rendezvous of contrack_amd/dist.py with two processes (the 128 id bytes travel through a file).
"""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np
import pytest
from scipy import ndimage

from contrack_amd import _native, dist as cdist

S2 = np.ones((3, 3), dtype=np.int32)
S3 = np.zeros((3, 3, 3), dtype=np.int32)
S3[1] = 1
S3[0, 1, 1] = S3[2, 1, 1] = 1                                        # contrack.py:748-750: 8-conn in plane + same pixel at t-1 / t+1


def _local_labelling(mask, t0, t1):
    """what one rank computes: 2-D components of the halo plane (t0-1) and of its own planes, union-find over them with the
    halo's components as the smallest indices, own roots ranked in raster order.
    Returns (last codes, halo codes, nroots, per own plane: array comp -> (kind, value))."""
    has_prev = t0 > 0
    planes = list(range(t0 - 1 if has_prev else t0, t1))
    labs, ncs = [], []
    for t in planes:
        lab, n = ndimage.label(mask[t], structure=S2)
        labs.append(lab)
        ncs.append(n)
    base = np.concatenate([[0], np.cumsum(ncs)])
    N = int(base[-1])
    par = np.arange(N)

    def find(i):
        while par[i] != i:
            par[i] = par[par[i]]
            i = par[i]
        return i
    for k in range(1, len(planes)):
        both = (labs[k] > 0) & (labs[k - 1] > 0)
        for a, b in set(zip(labs[k][both].tolist(), labs[k - 1][both].tolist())):
            ra, rb = find(base[k] + a - 1), find(base[k - 1] + b - 1)
            if ra != rb:
                par[max(ra, rb)] = min(ra, rb)
    nh = ncs[0] if has_prev else 0
    roots = [find(i) for i in range(N)]
    own_roots = sorted({r for r in roots if r >= nh})                # global component order = raster order of the first pixel
    rank_of = {r: k for k, r in enumerate(own_roots)}
    code = lambda r: (2 * r + 1) if r < nh else 2 * rank_of[r]      # noqa: E731
    last = [code(roots[base[-2] + c]) for c in range(ncs[-1])]
    halo = [roots[h] for h in range(nh)]
    per_plane = []
    for k in range(1 if has_prev else 0, len(planes)):
        per_plane.append([roots[base[k] + c] for c in range(ncs[k])])
    return last, halo, len(own_roots), (labs[1:] if has_prev else labs), per_plane, nh, rank_of


def _resolve(recs):
    world = len(recs)
    nlast = np.array([len(r[0]) for r in recs], dtype=np.int32)
    nh = np.array([len(r[1]) for r in recs], dtype=np.int32)
    nroots = np.array([r[2] for r in recs], dtype=np.int32)
    last = np.array(sum((r[0] for r in recs), []), dtype=np.int32)
    halo = np.array(sum((r[1] for r in recs), []), dtype=np.int32)
    off = np.zeros(world + 1, dtype=np.int64)
    ll = np.zeros(max(len(last), 1), dtype=np.int32)
    hl = np.zeros(max(len(halo), 1), dtype=np.int32)
    na = np.zeros(world, dtype=np.int32)
    last = np.ascontiguousarray(np.concatenate([last, [0]]).astype(np.int32))
    halo = np.ascontiguousarray(np.concatenate([halo, [0]]).astype(np.int32))
    _native.check(_native.lib().ctk_debug_boundary_resolve(world, nlast.ctypes.data, nh.ctypes.data, nroots.ctypes.data, last.ctypes.data,
                                                           halo.ctypes.data, off.ctypes.data, ll.ctypes.data, hl.ctypes.data, na.ctypes.data))
    return off, ll, hl, na, nlast, nh


@pytest.mark.parametrize("seed", range(40))
def test_boundary_resolve_gives_scipy_ids(seed):
    rng = np.random.default_rng(seed)
    T, ny, nx = int(rng.integers(2, 14)), int(rng.integers(3, 12)), int(rng.integers(3, 14))
    dens = float(rng.choice([0.2, 0.35, 0.5]))
    mask = rng.random((T, ny, nx)) < dens
    if rng.random() < 0.5:                                           # temporal persistence: blobs that live across cuts
        mask = np.repeat(mask[::2], 2, axis=0)[:T]
    want, nwant = ndimage.label(mask, structure=S3)
    world = int(rng.integers(1, min(T, 6) + 1))
    cuts = [0] + sorted(rng.choice(np.arange(1, T), size=world - 1, replace=False).tolist()) + [T]
    recs = [_local_labelling(mask, cuts[r], cuts[r + 1]) for r in range(world)]
    off, ll, hl, na, nlast, nh = _resolve(recs)
    assert off[-1] == nwant
    lo = np.concatenate([[0], np.cumsum(nlast)])
    ho = np.concatenate([[0], np.cumsum(nh)])
    # rebuild every pixel's id from the local roots the way k_rs_labels_sh does, and compare with scipy's
    for r in range(world):
        last, halo, nroots, labs, per_plane, nhr, rank_of = recs[r]
        hlab = hl[ho[r]:ho[r + 1]]
        llab = ll[lo[r]:lo[r + 1]]
        # absorbed own roots: visible through the last timestep's records
        absorbed = {}
        for c, v in enumerate(last):
            if v >= 0 and not (v & 1):
                absorbed.setdefault(v >> 1, llab[c])
        inv_rank = {k: root for root, k in rank_of.items()}
        true_label = {}
        abs_sorted = []
        # a root is absorbed iff the label its last-timestep component got differs from what its own rank would give
        for k in sorted(absorbed):
            own = off[r] + k - len(abs_sorted) + 1
            if absorbed[k] != own:
                abs_sorted.append(k)
        for k in range(nroots):
            before = sum(1 for a in abs_sorted if a < k)
            true_label[k] = absorbed[k] if k in abs_sorted else off[r] + k - before + 1
        assert len(abs_sorted) == na[r]
        for kp, lab in enumerate(labs):
            t = cuts[r] + kp
            got = np.zeros_like(lab)
            for c, root in enumerate(per_plane[kp]):
                got[lab == c + 1] = hlab[root] if root < nhr else true_label[rank_of[root]]
            assert np.array_equal(got, want[t]), (seed, r, t)


def test_boundary_resolve_rejects_contradictions():
    # rank 1 keeps a halo component whose twin on rank 0 was filtered out
    with pytest.raises(ValueError):
        _resolve([([-1], [], 0), ([0], [0], 1)])


def _rdzv_worker(rank, path, q):
    os.environ["CTK_RDZV_FILE"] = path
    payload = bytes(range(128))
    got = cdist.broadcast_bytes(rank, lambda: payload, 128, timeout_s=60.0)
    q.put((rank, got == payload))
    # rank 0 removes the file once every rank is known to have read it (in a real job: when the communicator exists)
    import time
    if rank == 0:
        t0 = time.time()
        while not os.path.exists(path + ".ack") and time.time() - t0 < 60:
            time.sleep(0.01)
        cdist.rendezvous_done(0, path)
    else:
        open(path + ".ack", "w").close()


def test_rendezvous_two_processes(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    path = str(tmp_path / "rdzv")
    procs = [ctx.Process(target=_rdzv_worker, args=(r, path, q)) for r in (1, 0)]      # the reader starts first
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert res == [(0, True), (1, True)]
    assert not os.path.exists(path)


def test_rendezvous_ignores_a_dead_writers_file(tmp_path):
    """a record left behind by a rank 0 that no longer exists (killed earlier launch, same key) must not be taken for the new one"""
    import struct
    import subprocess
    import sys
    from contrack_amd import _native
    path = str(tmp_path / "rdzv")
    p = subprocess.Popen([sys.executable, "-c", "pass"])
    p.wait()                                                   # a pid that existed and is gone
    with open(path, "wb") as f:
        f.write(b"CTKRDZV2" + struct.pack("<iiq", 0, p.pid, 8) + b"stale!!!")
    with pytest.raises(_native.CommError):
        cdist.broadcast_bytes(1, None, 8, path=path, timeout_s=0.5)
    # the live writer's record replaces it and is accepted
    assert cdist.broadcast_bytes(0, lambda: b"fresh..!", 8, path=path) == b"fresh..!"
    assert cdist.broadcast_bytes(1, None, 8, path=path, timeout_s=5.0) == b"fresh..!"


def test_rendezvous_publishes_rank0_failure(tmp_path):
    """rank 0 cannot make the id (RCCL does not start): the others hear about it at once instead of waiting for the deadline"""
    from contrack_amd import _native
    path = str(tmp_path / "rdzv")

    def boom():
        raise RuntimeError("librccl.so not found")
    with pytest.raises(RuntimeError):
        cdist.broadcast_bytes(0, boom, 128, path=path)
    with pytest.raises(_native.CommError, match="librccl.so not found"):
        cdist.broadcast_bytes(1, None, 128, path=path, timeout_s=30.0)


def test_rendezvous_names_are_per_communicator(monkeypatch):
    monkeypatch.delenv("CTK_RDZV_FILE", raising=False)
    assert cdist.rendezvous_file("id", 0) != cdist.rendezvous_file("id", 1)


def test_shard_bounds():
    assert cdist.shard_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]
    b = cdist.shard_bounds(2707, 8)
    assert b[0][0] == 0 and b[-1][1] == 2707 and all(x[1] == y[0] for x, y in zip(b, b[1:])) and all(y > x for x, y in b)


def test_dist_module_is_torch_free():
    """the product's multi-GPU driver binds the C ABI only: no torch import anywhere in the package"""
    import ast
    import glob
    pkg = os.path.dirname(os.path.abspath(cdist.__file__))
    for path in glob.glob(os.path.join(pkg, "*.py")):
        for node in ast.walk(ast.parse(open(path).read())):
            names = [a.name for a in node.names] if isinstance(node, ast.Import) else ([node.module or ""] if isinstance(node, ast.ImportFrom) else [])
            assert not any(n.split(".")[0] == "torch" for n in names), path


def test_result_pool_recycles_only_memory_nobody_holds():
    """the binding's pool of result memory (contrack_amd/_native._ResultPool), GPU-free: a block returns to the pool when the LAST
    view of the array handed out is gone, is handed out again for a result of the same size, and the pool never holds more than
    its cap"""
    import gc
    from contrack_amd import _native

    class FakeTracker:                       # (no handle: the pool then never registers anything with HIP)
        handle = None
    trk = FakeTracker()
    pool = _native._ResultPool(trk)
    shape = (40, 181, 360)                   # 10.4 MB: above the pool's threshold
    a = pool.take(shape)
    assert a.shape == shape and a.dtype == np.int32 and a.flags.writeable and a.flags.c_contiguous
    a[...] = 7
    addr = a.ctypes.data
    v = a[3:5, 10]                           # a view keeps the block leased
    del a
    gc.collect()
    b = pool.take(shape)
    assert b.ctypes.data != addr and pool.hits == 0 and int(v[0, 0]) == 7
    del v
    gc.collect()
    c = pool.take(shape)                     # now the first block is free again
    assert c.ctypes.data == addr and pool.hits == 1
    small = pool.take((4, 5, 6))             # small results are plain arrays
    assert small.base is None or small.nbytes < (8 << 20)
    pool.cap = b.nbytes                      # room for one block only
    del b, c
    gc.collect()
    assert sum(x["mem"].nbytes for x in pool._free) <= pool.cap
    pool.close()
    assert pool._free == []
