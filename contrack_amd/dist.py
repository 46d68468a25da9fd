"""Time-sharded multi-GPU driver of the run_contrack hot path -- no torch.

One process per GPU (any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT, e.g. torch.distributed.run as
the driver's launch contract prescribes -- the launcher is the only thing taken from it); rank r owns the contiguous timesteps
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
import os
import struct
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
