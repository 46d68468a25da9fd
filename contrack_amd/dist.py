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
staging through POSIX shared memory: RCCL refuses two ranks on one device; plumbing tests on a one-GPU box).
"""
import atexit
import ctypes as C
import json
import os
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


def rendezvous_file(tag="id"):
    return os.environ.get("CTK_RDZV_FILE") or os.path.join(os.environ.get("TMPDIR", "/tmp"), "ctk_rdzv_%s_%s" % (launch_key(), tag))


def broadcast_bytes(rank, make, nbytes, path=None, timeout_s=180.0):
    """rank 0 calls make() -> bytes of length nbytes and publishes them through a file; the other ranks wait for it.
    Single node (the ranks share a file system); the file is removed by rank 0 at exit."""
    path = path or rendezvous_file()
    if rank == 0:
        data = make()
        assert len(data) == nbytes
        tmp = "%s.tmp.%d" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, path)                       # atomic: readers see nothing or everything

        def _cleanup(p=path):
            try:
                os.remove(p)
            except OSError:
                pass
        atexit.register(_cleanup)
        return data
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                data = f.read()
            if len(data) == nbytes:
                return data
        except OSError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError("rank %d: rendezvous file %s did not appear within %.0f s" % (rank, path, timeout_s))
        time.sleep(0.01)


def env_rank_world():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", rank))
    return rank, world, local


def init_comm(tracker, rank, world, backend=None):
    """the communicator of this rank: RCCL (default) or the shared-memory transport"""
    backend = backend or os.environ.get("CTK_DIST_BACKEND", "rccl")
    if backend == "rccl":
        def make_id():
            try:
                return _native.comm_unique_id()
            except _native.ContrackHipError as e:          # librccl.so cannot be loaded / refuses to start on this node
                sys.stderr.write("contrack_amd.dist: RCCL is not usable here (%s); all ranks use the shared-memory transport\n" % e)
                return bytes(_native.COMM_ID_BYTES)         # the all-zero id tells every rank the same thing
        uid = broadcast_bytes(rank, make_id, _native.COMM_ID_BYTES)
        if uid != bytes(_native.COMM_ID_BYTES):
            c = _native.Comm.rccl(tracker, uid, rank, world)
            c.transport = "rccl"
            return c
        backend = "shm"
    if backend == "shm":
        c = _native.Comm.shm(tracker, "ctk_%s" % launch_key(), rank, world)
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
        self.comm = init_comm(self.trk, self.rank, self.world, backend)

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        self.trk.close()

    def track(self, anom_local, t_begin, T_total, thr_local, cmp_op, wrow, overlap, persistence, twosided=True):
        a = np.ascontiguousarray(anom_local)
        f64 = a.dtype != np.float32
        if f64:
            a = a.astype(np.float64)
        T, ny, nx = a.shape
        d_in, d_out = self.trk.malloc(max(a.nbytes, 8)), self.trk.malloc(max(T * ny * nx * 4, 8))
        try:
            self.trk.h2d(d_in, a)
            n = self.trk.track_sharded_dev(self.comm, d_in, T, t_begin, T_total, ny, nx, thr_local, cmp_op, wrow, overlap, persistence,
                                           twosided, d_out, f64=f64)
            flag = np.empty((T, ny, nx), dtype=np.int32)
            self.trk.d2h(flag, d_out)
        finally:
            self.trk.free(d_in)
            self.trk.free(d_out)
        return flag, n


# ------------------------------------------------------------------------------------------------
# bench.py leg for N > 1 (one rank per GPU, launched as the bench contract says)
# ------------------------------------------------------------------------------------------------
def bench_main(args, wl, workloads, hbm_peak):
    from . import synth
    # Keep stdout clean for the ONE JSON line: RCCL may print through C stdio at communicator creation.  Everything
    # written to fd 1 until the result is ready goes to stderr instead.
    sys_stdout_fd = os.dup(1)
    os.dup2(2, 1)
    rank, world, local = env_rank_world()
    backend = os.environ.get("CTK_DIST_BACKEND", "rccl")
    st = ShardedTracker(rank=rank, world=world, backend=backend)
    trk, comm = st.trk, st.comm
    backend = getattr(comm, "transport", backend)              # (shm if RCCL could not be loaded)
    T, ny, nx = wl["T"], wl["ny"], wl["nx"]
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
    d_in = trk.malloc(max(nloc * ny * nx * 4, 8))
    d_out = trk.malloc(max(nloc * ny * nx * 4, 8))
    if wl.get("device_fill"):
        trk.synth_fill(d_in, nloc, ny, nx, seed=rank if weak else 0)       # (strong + device_fill: every rank its own window)
    elif weak:
        trk.h2d(d_in, synth.smooth_field(T, ny, nx, seed=rank))
    else:
        trk.h2d(d_in, synth.smooth_field(T, ny, nx, seed=0)[t0:t1])
    lat, _ = synth.grid(ny, nx)
    w = np.array((111 * np.float32(180.0 / (ny - 1)) * 111 * np.float32(360.0 / nx) * np.cos(lat * np.pi / 180))).astype(np.float32)
    thr = np.full(nloc, np.float64(np.float32(wl["threshold"])))
    op = _native.CMP_OPS[wl["gorl"]]
    trk.set_timing(1)            # HIP events around the two streaming kernels only (see bench.py)

    def step():
        return trk.track_sharded_dev(comm, d_in, nloc, t0, T_total, ny, nx, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"], d_out)

    n_tracked = None
    for _ in range(max(args.warmup, 0)):
        n_tracked = step()
    comm.barrier()
    trk.sync()
    acc = {}
    ops0 = comm.ops()
    tb = time.perf_counter()
    for _ in range(args.steps):
        n_tracked = step()
        for k, v in trk.timings().items():
            acc[k] = acc.get(k, 0.0) + v
    trk.sync()
    comm.barrier()
    dt_local = time.perf_counter() - tb
    dt = float(comm.allgather(np.array([dt_local], dtype=np.float64)).max())
    ops1 = comm.ops()
    per = {k: v / max(args.steps, 1) for k, v in acc.items()}
    stats = trk.stats()
    px = nloc * ny * nx
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
                       collectives_per_step=dict(neighbour_exchanges=(ops1["neighbour_exchanges"] - ops0["neighbour_exchanges"]) / nsteps,
                                                 allgathers=(ops1["allgathers"] - ops0["allgathers"]) / nsteps)),
                   roofline=dict(bound="hbm", kernel={"k_threshold": "k_threshold_v4", "k_relabel": {5: "k_relabel_v5", 4: "k_relabel_v4"}.get(stats.get("relabel_kernel", 4), "k_relabel")}[kern], achieved=achieved, peak=hbm_peak,
                                 unit="GB/s", frac=achieved / hbm_peak, traffic=None, algorithmic_bytes_per_launch=alg[kern],
                                 avg_kernel_ms=per.get(kern), note="rank 0's shard"),
                   kernels_ms=per, workload_stats_rank0=stats)
        try:
            C.CDLL(None).fflush(None)            # drain C stdio into stderr before stdout is restored
        except Exception:
            pass
        os.dup2(sys_stdout_fd, 1)
        print(json.dumps(out), flush=True)
    trk.free(d_in)
    trk.free(d_out)
    st.close()
