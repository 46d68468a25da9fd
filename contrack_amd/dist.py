"""Time-sharded multi-GPU driver of the run_contrack hot path -- no torch.

One process per GPU (any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT, e.g. torch.distributed.run as
bench.py's contract prescribes -- the launcher is the only thing taken from it); rank r owns the contiguous timesteps
[t0_r, t1_r).  Everything that follows happens inside libcontrack_hip.so (csrc/ctk_sharded.hip, ctk_comm.hip): the
ranks exchange a one-timestep label-map halo with their neighbours and a few hundred boundary records per step
through RCCL (ncclSend/ncclRecv, ncclAllGather over xGMI); the overlap filter, the 3-D labelling and the seam merges are
resolved shard-locally with those boundary conditions.  Bulk pixel data never crosses the fabric.

What Python does here: split the time axis, bring the ranks together (the 128-byte ncclUniqueId travels through a file
in /tmp keyed by the launcher -- single node, no sockets, no torch), and call ctk_track_sharded_*.

Transports (CTK_DIST_BACKEND): "rccl" (default, one GPU per rank), "shm" (several processes sharing ONE GPU, host
staging through POSIX shared memory: RCCL refuses two ranks on one device; plumbing tests on a one-GPU box).  There is no
silent fallback: if RCCL cannot start, every rank raises.

Failure behaviour: a rank that fails tells the others through the communicator's control segment (csrc/ctk_comm.h); their calls
raise _native.CommError within milliseconds.  Dead ranks are noticed by pid; everything else by a deadline
(CTK_COMM_TIMEOUT_S, default 120 s).
"""
import ctypes as C
import json
import os
import struct
import sys
import time

import numpy as np

# the host driver of the target nodes only supports dmabuf IPC: without this RCCL's intra-node set-up fails with
# `hipIpcGetMemHandle: invalid argument` (read by the HIP runtime when it initialises, i.e. before the first library call)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from . import _native  # noqa: E402


def shard_bounds(T, world):
    """Contiguous, balanced split of T timesteps over `world` ranks: list of (t0, t1)."""
    base, rem = divmod(int(T), int(world))
    out, t = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((t, t + n))
        t += n
    return out


# ------------------------------------------------------------------------------------------------
# bringing the ranks together
# ------------------------------------------------------------------------------------------------
def launch_key():
    """what the ranks of ONE launch share and other launches on this node do not"""
    return "%s_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                         int(os.environ.get("CTK_LAUNCH_PID", os.getppid())))


# communicators created by this process so far: every rank of a launch creates them in the same order, so the number names
# the rendezvous of ONE communicator (a second ShardedTracker of the same launch never reads the first one's file)
_comm_seq = [0]


def rendezvous_file(tag="id", seq=None):
    if os.environ.get("CTK_RDZV_FILE"):
        return os.environ["CTK_RDZV_FILE"]
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "ctk_rdzv_%s_%d_%s" % (launch_key(), _comm_seq[0] if seq is None else seq, tag))


_RDZV_MAGIC = b"CTKRDZV2"


def _pid_alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    try:
        with open("/proc/%d/stat" % pid) as f:
            st = f.read()
        return st[st.rindex(")") + 2] not in "ZX"          # killed but not reaped yet
    except (OSError, ValueError, IndexError):
        return True


def broadcast_bytes(rank, make, nbytes, path=None, timeout_s=None):
    """rank 0 calls make() -> bytes of length nbytes and publishes them through a file; the other ranks wait for it.
    Single node (the ranks share a file system).  The record carries rank 0's pid: a file left behind by a rank 0 that no
    longer exists (a killed earlier launch with the same key) is ignored; a make() that raises is published as such, so that
    the other ranks fail at once with rank 0's message instead of waiting for the deadline.  Remove the file with
    rendezvous_done() once every rank is known to have read it (i.e. after the communicator exists)."""
    path = path or rendezvous_file()
    timeout_s = float(os.environ.get("CTK_COMM_TIMEOUT_S", "120")) if timeout_s is None else timeout_s
    if rank == 0:
        err = None
        try:
            data = make()
            assert len(data) == nbytes
            rec = _RDZV_MAGIC + struct.pack("<iiq", 0, os.getpid(), nbytes) + data
        except BaseException as e:      # noqa: BLE001 -- told to the other ranks, then re-raised
            err = e
            msg = ("%s: %s" % (type(e).__name__, e)).encode("utf-8", "replace")[:2000]
            rec = _RDZV_MAGIC + struct.pack("<iiq", 1, os.getpid(), len(msg)) + msg
        tmp = "%s.tmp.%d" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(rec)
        os.replace(tmp, path)                       # atomic: readers see nothing or everything
        if err is not None:
            raise err
        return data
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                rec = f.read()
            if len(rec) >= 24 and rec[:8] == _RDZV_MAGIC:
                status, pid, n = struct.unpack("<iiq", rec[8:24])
                if len(rec) == 24 + n and _pid_alive(pid):
                    if status != 0:
                        raise _native.CommError("rank %d: rank 0 could not start the communicator: %s" % (rank, rec[24:].decode("utf-8", "replace")))
                    if n == nbytes:
                        return rec[24:]
        except OSError:
            pass
        if time.time() - t0 > timeout_s:
            raise _native.CommError("rank %d: rendezvous file %s did not appear within %.0f s" % (rank, path, timeout_s))
        time.sleep(0.005)


def rendezvous_done(rank, path):
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass


def env_rank_world():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", rank))
    return rank, world, local


def init_comm(tracker, rank, world, backend=None):
    """the communicator of this rank: RCCL (default) or the shared-memory transport (CTK_DIST_BACKEND=shm).  No fallback from
    one to the other: a node where RCCL does not start is a node where the multi-GPU path does not run, and says so."""
    backend = backend or os.environ.get("CTK_DIST_BACKEND", "rccl")
    seq = _comm_seq[0]
    _comm_seq[0] += 1
    if backend == "rccl":
        path = rendezvous_file("id", seq)
        uid = broadcast_bytes(rank, _native.comm_unique_id, _native.COMM_ID_BYTES, path)
        try:
            c = _native.Comm.rccl(tracker, uid, rank, world)         # (returns once every rank has arrived)
        finally:
            rendezvous_done(rank, path)
        c.transport = "rccl"
        return c
    if backend == "shm":
        path = rendezvous_file("shm", seq)
        nonce = broadcast_bytes(rank, lambda: os.urandom(8), 8, path)      # a segment name nobody has used before
        try:
            c = _native.Comm.shm(tracker, "ctk_%s" % nonce.hex(), rank, world)
        finally:
            rendezvous_done(rank, path)
        c.transport = "shm"
        return c
    raise ValueError("CTK_DIST_BACKEND must be 'rccl' or 'shm', not %r" % backend)


class ShardedTracker:
    """run_contrack on one time shard per rank.  `track` takes this rank's slice of the (time, lat, lon) slab as a host
    array and returns its slice of `flag` and the number of tracked contours (identical on every rank)."""

    def __init__(self, device=None, rank=None, world=None, backend=None):
        r, w, local = env_rank_world()
        self.rank = r if rank is None else int(rank)
        self.world = w if world is None else int(world)
        backend = backend or os.environ.get("CTK_DIST_BACKEND", "rccl")
        ndev = max(1, _native.device_count())
        # RCCL wants one device per rank; the shared-memory transport also runs several ranks on one device (tests on a 1-GPU box)
        self.device = (local if backend == "rccl" else local % ndev) if device is None else int(device)
        self.trk = _native.Tracker(self.device)
        self.comm = None
        try:
            self.comm = init_comm(self.trk, self.rank, self.world, backend)
        except BaseException:
            self.trk.close()
            raise

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        self.trk.close()

    def track(self, anom_local, t_begin, T_total, thr_local, cmp_op, wrow, overlap, persistence, twosided=True):
        """Collective.  A rank that fails BEFORE it reaches the C entry (bad arguments, no device memory, a failed upload) aborts the
        communicator, so that the other ranks raise CommError instead of waiting for it until the deadline; inside the entry the
        library does that itself -- except for errors every rank derives from the same gathered data, which leave the communicator
        usable (csrc/ctk_sharded.hip, COLLECTIVE_FAIL)."""
        d_in = d_out = None
        try:
            try:
                a = np.ascontiguousarray(anom_local)
                f64 = a.dtype != np.float32
                if f64:
                    a = a.astype(np.float64)
                if a.ndim != 3:
                    raise ValueError("the local slab must be (time, lat, lon)")
                T, ny, nx = a.shape
                if np.shape(thr_local) != (T,):
                    raise ValueError("thr_local must hold one value per local timestep")
                if np.shape(wrow) != (ny,):
                    raise ValueError("wrow must hold one weight per latitude row")
                d_in = self.trk.malloc(max(a.nbytes, 8))
                d_out = self.trk.malloc(max(T * ny * nx * 4, 8))
                self.trk.h2d(d_in, a)
            except BaseException:
                try:
                    self.comm.abort(-5)              # the other ranks must not wait for this one
                except Exception:                    # noqa: BLE001 -- the original error is the one to report
                    pass
                raise
            n = self.trk.track_sharded_dev(self.comm, d_in, T, t_begin, T_total, ny, nx, thr_local, cmp_op, wrow, overlap, persistence,
                                           twosided, d_out, f64=f64)
            flag = np.empty((T, ny, nx), dtype=np.int32)
            self.trk.d2h(flag, d_out)
        finally:
            for p in (d_in, d_out):
                if p is not None:
                    self.trk.free(p)
        return flag, n


# ------------------------------------------------------------------------------------------------
# bench.py leg for N > 1 (one rank per GPU, launched as the bench contract says)
# ------------------------------------------------------------------------------------------------
def _ptr(p, off):
    return C.c_void_p((p.value if hasattr(p, "value") else int(p)) + int(off))


def _scratch_dir():
    d = os.environ.get("CTK_BENCH_TMP") or ("/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else os.environ.get("TMPDIR", "/tmp"))
    return d


def _weights(ny, nx):
    from . import synth
    lat, _ = synth.grid(ny, nx)
    return np.array((111 * np.float32(180.0 / (ny - 1)) * 111 * np.float32(360.0 / nx) * np.cos(lat * np.pi / 180))).astype(np.float32)


def _timed_sharded(trk, comm, step, steps, warmup):
    """barrier + sync | `steps` calls | sync + barrier; max over ranks (the bench contract's timing)"""
    n = None
    for _ in range(max(warmup, 0)):
        n = step()
    comm.barrier()
    trk.sync()
    ops0 = comm.ops()
    tb = time.perf_counter()
    for _ in range(steps):
        n = step()
    trk.sync()
    comm.barrier()
    dt_local = time.perf_counter() - tb
    dt = float(comm.allgather(np.array([dt_local], dtype=np.float64)).max())
    ops1 = comm.ops()
    k = max(steps, 1)
    return n, dt, dict(neighbour_exchanges=(ops1["neighbour_exchanges"] - ops0["neighbour_exchanges"]) / k,
                       allgathers=(ops1["allgathers"] - ops0["allgathers"] - 2) / k)       # (- the two barriers / the time gather)


def parity_check(trk, comm, rank, world, members, d_out_local, nloc, t0, T_total, ny, nx, thr_value, op, w, wl, n_tracked):
    """In-run proof that the sharded result is the one-call result: every rank checksums its shard of `flag` on its GPU; rank 0
    tracks the concatenated slab with ONE call on its own GPU and checksums the same windows.  `members`: how rank 0 obtains
    member q's input -- ("file", path) written by rank q, or ("fill", seed, t_first, T_q) for the device generator.
    Returns (checked, detail) on rank 0, (None, None) elsewhere.  Collective: every rank must call."""
    from . import synth
    plane = ny * nx
    mine = np.array(trk.checksum_i32(d_out_local, nloc * plane, t0 * plane), dtype=np.uint64)
    allsum = comm.allgather(mine)                                     # (world, 2)
    bounds = comm.allgather(np.array([t0, nloc], dtype=np.int64))
    detail = None
    if rank == 0:
        d_in = d_out = None
        try:
            d_in = trk.malloc(T_total * plane * 4)
            d_out = trk.malloc(T_total * plane * 4)
            for q in range(world):
                tq, nq = int(bounds[q][0]), int(bounds[q][1])
                m = members[q]
                if m[0] == "file":
                    trk.h2d(_ptr(d_in, tq * plane * 4), np.load(m[1], mmap_mode="r"))
                elif m[0] == "regen":                                 # (no shared scratch space: rank 0 generates the member again)
                    trk.h2d(_ptr(d_in, tq * plane * 4), np.ascontiguousarray(synth.smooth_field(m[2], ny, nx, seed=m[1])[m[3]:m[3] + nq]))
                else:
                    trk.synth_fill(_ptr(d_in, tq * plane * 4), nq, ny, nx, seed=m[1], t0=m[2])
            thr = np.full(T_total, thr_value)
            trk.set_timing(0)
            n_one = trk.track_dev(d_in, T_total, ny, nx, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"], d_out)
            equal = []
            for q in range(world):
                tq, nq = int(bounds[q][0]), int(bounds[q][1])
                ref = trk.checksum_i32(_ptr(d_out, tq * plane * 4), nq * plane, tq * plane)
                equal.append(bool(ref[0] == int(allsum[q][0]) and ref[1] == int(allsum[q][1])))
            detail = dict(shards_equal=equal, n_tracked_one_call=int(n_one), n_tracked_sharded=int(n_tracked),
                          flag_pixels=int(allsum[:, 1].sum()),
                          method="64-bit position-weighted checksum of every rank's flag shard vs the same window of ONE ctk_track_f32_dev "
                                 "call on the concatenated %dx%dx%d slab on rank 0's GPU" % (T_total, ny, nx))
        except (MemoryError, OSError, _native.ContrackHipError, ValueError) as e:
            detail = dict(error="%s: %s" % (type(e).__name__, e))
        finally:
            for p in (d_in, d_out):
                if p is not None:
                    trk.free(p)
    comm.barrier()
    if rank != 0:
        return None, None
    ok = bool(detail.get("shards_equal")) and all(detail["shards_equal"]) and detail["n_tracked_one_call"] == detail["n_tracked_sharded"]
    return ok, detail


def strong_block(trk, comm, rank, world, wl, T, steps, warmup):
    """Strong scaling of ONE device-generated slab of T steps (the north_star's >= 6x target is stated on 0.25 deg): rank 0 times
    the one-call path on the whole slab on its GPU, then all ranks track their windows of the same slab; speed-up = the ratio."""
    ny, nx = wl["ny"], wl["nx"]
    plane = ny * nx
    w = _weights(ny, nx)
    op = _native.CMP_OPS[wl["gorl"]]
    thr_value = np.float64(np.float32(wl["threshold"]))
    one = np.zeros(2, dtype=np.float64)                                # ms per pass on one GPU, tracked count
    err = None
    if rank == 0:
        d_in = d_out = None
        try:
            d_in, d_out = trk.malloc(T * plane * 4), trk.malloc(T * plane * 4)
            trk.synth_fill(d_in, T, ny, nx, seed=0)
            thr = np.full(T, thr_value)
            trk.set_timing(0)
            for _ in range(max(warmup, 1)):
                n1 = trk.track_dev(d_in, T, ny, nx, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"], d_out)
            trk.sync()
            tb = time.perf_counter()
            for _ in range(steps):
                n1 = trk.track_dev(d_in, T, ny, nx, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"], d_out)
            trk.sync()
            one[:] = ((time.perf_counter() - tb) * 1e3 / steps, n1)
        except (MemoryError, _native.ContrackHipError, ValueError) as e:
            err = "%s: %s" % (type(e).__name__, e)
        finally:
            for p in (d_in, d_out):
                if p is not None:
                    trk.free(p)
    one = comm.allgather(one)[0]
    if one[0] <= 0:                                                    # (every rank sees it: the caller may try a smaller slab)
        return False, (dict(error=err or "the one-GPU pass did not run", steps_tried=T) if rank == 0 else None)
    t0, t1 = shard_bounds(T, world)[rank]
    nloc = t1 - t0
    d_in, d_out = trk.malloc(nloc * plane * 4), trk.malloc(nloc * plane * 4)
    trk.synth_fill(d_in, nloc, ny, nx, seed=0, t0=t0)
    thr = np.full(nloc, thr_value)

    def step():
        return trk.track_sharded_dev(comm, d_in, nloc, t0, T, ny, nx, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"], d_out)
    nN, dt, coll = _timed_sharded(trk, comm, step, steps, warmup)
    trk.free(d_in)
    trk.free(d_out)
    if rank != 0:
        return True, None
    ms = dt * 1e3 / steps
    return True, dict(workload="%dx%dx%d float32 (device-generated), threshold %s %g, overlap %g, persistence %d" % (
                    T, ny, nx, wl["gorl"], wl["threshold"], wl["overlap"], wl["persistence"]),
                n_gpus=world, ms_per_step_1gpu=float(one[0]), ms_per_step=ms, speedup_vs_1gpu=float(one[0]) / ms,
                timesteps_per_s=T / (ms * 1e-3), n_tracked=int(nN), n_tracked_one_call=int(one[1]), n_tracked_equal=bool(int(one[1]) == int(nN)),
                collectives_per_step=coll, steps=steps,
                note="same launch: the one-call time is measured on rank 0's GPU (the whole slab resident), then the slab is split into "
                     "%d time shards; barrier + sync around the timed calls, max over ranks" % world)


def bench_main(args, wl, workloads, hbm_peak, cpu_baseline=None, pmc_traffic=None):
    from . import synth
    # Keep stdout clean for the ONE JSON line: RCCL may print through C stdio at communicator creation.  Everything
    # written to fd 1 until the result is ready goes to stderr instead.
    sys_stdout_fd = os.dup(1)
    os.dup2(2, 1)
    rank, world, local = env_rank_world()
    backend = os.environ.get("CTK_DIST_BACKEND", "rccl")
    st = ShardedTracker(rank=rank, world=world, backend=backend)
    trk, comm = st.trk, st.comm
    backend = getattr(comm, "transport", backend)
    T, ny, nx = wl["T"], wl["ny"], wl["nx"]
    plane = ny * nx
    weak = getattr(args, "scaling", "weak") == "weak"
    if weak:
        # weak scaling: one member of wl["T"] steps per GPU, members concatenated on the time axis (the layout of
        # BASELINE.json configs[4]); rank r holds member r = synthetic slab with seed r, global steps [r T, (r+1) T).
        t0, t1 = rank * T, (rank + 1) * T
        T_total = T * world
    else:
        t0, t1 = shard_bounds(T, world)[rank]
        T_total = T
    nloc = t1 - t0
    d_in = trk.malloc(max(nloc * plane * 4, 8))
    d_out = trk.malloc(max(nloc * plane * 4, 8))
    check_parity = not getattr(args, "no_parity_check", False) and 2 * T_total * plane * 4 <= (96 << 30)
    member = None
    cpu_slab = None
    if wl.get("device_fill"):
        seed, tf = (rank, 0) if weak else (0, t0)
        trk.synth_fill(d_in, nloc, ny, nx, seed=seed, t0=tf)
        member = ("fill", seed, tf)
    else:
        a_full = synth.smooth_field(T, ny, nx, seed=rank if weak else 0)
        a = a_full if weak else a_full[t0:t1]
        trk.h2d(d_in, a)
        if rank == 0 and cpu_baseline is not None and not getattr(args, "no_cpu_baseline", False):
            cpu_slab = a_full                                          # (the workload's own slab, seed 0: what the N = 1 line times too)
        if check_parity:
            # rank 0 needs every member for the one-call run: through a file in shared scratch space, or -- if that cannot be
            # written -- by generating it again from (seed, window)
            member = ("regen", rank if weak else 0, T, 0 if weak else t0)
            for d in (_scratch_dir(), os.environ.get("TMPDIR", "/tmp")):
                path = os.path.join(d, "ctk_bench_%s_member%d.npy" % (launch_key(), rank))
                try:
                    np.save(path, a)
                    member = ("file", path)
                    break
                except OSError:
                    try:
                        os.remove(path)
                    except OSError:
                        pass
        del a, a_full
    w = _weights(ny, nx)
    thr_value = np.float64(np.float32(wl["threshold"]))
    thr = np.full(nloc, thr_value)
    op = _native.CMP_OPS[wl["gorl"]]
    trk.set_timing(1)            # HIP events around the two streaming kernels only (see bench.py)

    def step():
        return trk.track_sharded_dev(comm, d_in, nloc, t0, T_total, ny, nx, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"], d_out)

    n_tracked = None
    for _ in range(max(args.warmup, 0)):
        n_tracked = step()
    comm.barrier()
    trk.sync()
    trk.timing_sums(reset=True)  # (level-1 timing: an event pair around ONE of the two streaming kernels in every second pass, summed by the library)
    ops0 = comm.ops()
    tb = time.perf_counter()
    for _ in range(args.steps):
        n_tracked = step()
    trk.sync()
    comm.barrier()
    dt_local = time.perf_counter() - tb
    dt = float(comm.allgather(np.array([dt_local], dtype=np.float64)).max())
    ops1 = comm.ops()
    per, nmeas = trk.timing_sums(reset=True)
    last = trk.timings()                              # (host-side timers of the last pass: informational)
    per.update({k: last[k] for k in ("host_seam_driver", "total", "d2h", "h2d")})
    stats = trk.stats()
    px = nloc * plane

    # ---- untimed: the result proves itself (in-run parity against the one-call path), then the strong-scaling leg -----------
    parity_ok, parity = None, None
    if check_parity:
        code = {"fill": 0, "file": 1, "regen": 3}[member[0]]
        if member[0] == "file" and os.path.dirname(member[1]) != _scratch_dir():
            code = 2                                                  # (the file went to TMPDIR)
        kinds = comm.allgather(np.array([code] + [int(v) for v in member[1:4] if not isinstance(v, str)] + [0] * 3, dtype=np.int64)[:4])
        members = []
        for q in range(world):
            c, k = int(kinds[q][0]), [int(v) for v in kinds[q][1:4]]
            if c in (1, 2):
                members.append(("file", os.path.join(_scratch_dir() if c == 1 else os.environ.get("TMPDIR", "/tmp"), "ctk_bench_%s_member%d.npy" % (launch_key(), q))))
            elif c == 3:
                members.append(("regen", k[0], k[1], k[2]))
            else:
                members.append(("fill", k[0], k[1]))
        parity_ok, parity = parity_check(trk, comm, rank, world, members, d_out, nloc, t0, T_total, ny, nx, thr_value, op, w, wl, n_tracked)
        if member[0] == "file":
            try:
                os.remove(member[1])
            except OSError:
                pass
    trk.free(d_in)
    trk.free(d_out)
    strong = None
    sT = int(getattr(args, "strong_steps", 0) or 0)
    if world > 1 and sT != 0:
        swl = workloads["era5_025deg_2k"]                              # (grid and parameters; the number of steps is sT)
        tried = []
        for cand in ([14600, 2000] if sT < 0 else [sT]):
            if cand < world:
                continue
            ok, strong = strong_block(trk, comm, rank, world, swl, cand, max(2, min(args.steps, 5)), 1)
            if ok:
                break
            tried.append(strong)
        if rank == 0 and strong is not None and tried and "error" not in strong:
            strong["larger_slab_not_run"] = tried

    n_devices = len(set(int(v) for v in comm.allgather(np.array([st.device], dtype=np.int64)).ravel()))
    alg = {"k_threshold": 4.0 * px, "k_relabel": 4.0 * px}
    kern = max(alg, key=lambda k: per.get(k, 0.0))
    achieved = alg[kern] / (per[kern] * 1e-3) / 1e9 if per.get(kern, 0) > 0 else 0.0
    if rank == 0:
        nsteps = max(args.steps, 1)
        out = dict(metric="timesteps/sec labeled+tracked", value=T_total * args.steps / dt, unit="timesteps/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=dt * 1e3 / nsteps, higher_is_better=True,
                   scaling="weak" if weak else "strong", vs_baseline=None, dtype="f32 compare / int32 labels / int64 exact areas", data="synthetic",
                   config=dict(workload="%s: %s%dx%dx%d float32, threshold %s %g, overlap %g, persistence %d, twosided %s" % (
                       args.workload, ("%d members concatenated on the time axis, each " % world) if weak else "", T, ny, nx, wl["gorl"],
                       wl["threshold"], wl["overlap"], wl["persistence"], wl["twosided"]),
                       total_timesteps=T_total,
                       parallelism="time-sharded x%d: one-timestep halo + boundary records, shard-local resolver (%s)" % (
                           world, "RCCL ncclSend/Recv + ncclAllGather" if backend == "rccl" else "shared-memory transport (host staging; several ranks may share a GPU)"),
                       transport=backend, rccl_ranks=world if backend == "rccl" else 0, n_tracked=n_tracked,
                       parity_checked=bool(parity_ok) if parity_ok is not None else False, parity=parity,
                       collectives_per_step=dict(neighbour_exchanges=(ops1["neighbour_exchanges"] - ops0["neighbour_exchanges"]) / nsteps,
                                                 allgathers=(ops1["allgathers"] - ops0["allgathers"] - 2) / nsteps)),      # (- the closing barrier and the time gather)
                   roofline=dict(bound="hbm", kernel={"k_threshold": "k_threshold_v4" if os.environ.get("CTK_THRESHOLD") == "4" else "k_threshold_v7", "k_relabel": {5: "k_relabel_v5", 4: "k_relabel_v4"}.get(stats.get("relabel_kernel", 4), "k_relabel")}[kern], achieved=achieved, peak=hbm_peak,
                                 unit="GB/s", frac=achieved / hbm_peak, traffic=None, algorithmic_bytes_per_launch=alg[kern],
                                 avg_kernel_ms=per.get(kern), note="rank 0's shard"),
                   kernels_ms=per, workload_stats_rank0=stats)
        out["config"]["distinct_devices"] = n_devices
        if pmc_traffic is not None and nloc == T:
            # (HBM bytes per launch from the committed PMC capture of this workload: rank 0's shard IS the workload's slab when every
            # rank holds one member)
            out["roofline"]["traffic"] = pmc_traffic(out["roofline"]["kernel"], args.workload)
        if strong is not None:
            out["strong_025deg"] = strong
    st.close()                                       # (nothing collective from here on)
    rc = 0
    if rank == 0:
        if cpu_slab is not None:
            # the CPU leg: the oracle's scipy port on the workload's own slab, rank 0's host cores, after every collective is done
            out["cpu_baseline"] = cpu_baseline(wl, cpu_slab, w)
        # the line must not stand for a result nobody checked: a parity check that ran and failed (or could not run) fails the job
        if check_parity and not parity_ok:
            rc = 3
            print("bench.py: the in-run parity check of the sharded result FAILED or could not run: %s" % json.dumps(parity), file=sys.stderr)
        try:
            C.CDLL(None).fflush(None)            # drain C stdio into stderr before stdout is restored
        except Exception:
            pass
        os.dup2(sys_stdout_fd, 1)
        print(json.dumps(out), flush=True)
    return rc
