"""Time-sharded multi-GPU driver for the run_contrack hot path.

One process per GPU (torch.distributed.run); rank r owns the contiguous timesteps [t0_r, t1_r).

  stage 1  threshold + 2-D labelling + seam merge            independent per timestep, no communication
  halo     rank r sends its labelled LAST timestep (bit mask + run->component ids, the compressed
           one-timestep label map) to rank r+1                send/recv to the ring neighbour (RCCL over xGMI)
  stage 2  label co-occurrence histogram (t, t-1)             local; the first local step uses the halo
  tables   component / pair / seam tables of every shard      all-gather (small: ~40 B per component)
  resolve  overlap recurrence, 3-D ids, seam merges           replicated on every rank on the gathered tables
  extents  per-id time extents for the persistence filter     local, then all-reduce MIN / MAX (one int32 per id)
  write    relabel pass writes the rank's slice of `flag`     local

Bulk pixel data never crosses the fabric, so no ring all-reduce of slab-sized buffers appears.

The driver is written against two small interfaces so that its logic is testable without a GPU:
  engine  -- the per-shard stages (HipShardEngine below wraps libcontrack_hip.so; tests/ has a numpy one)
  comm    -- TorchComm over torch.distributed: backend "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU.
"""
import ctypes as C
import json
import os
import time

import numpy as np

from . import _native


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
# communicator
# ------------------------------------------------------------------------------------------------
class _DevArray:
    """Zero-copy view of raw device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class TorchComm:
    """torch.distributed plumbing.  device=None -> CPU tensors (gloo); else CUDA/HIP tensors (nccl = RCCL)."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device

    def _dev(self):
        return self.torch.device("cpu") if self.device is None else self.torch.device("cuda", self.device)

    def barrier(self):
        if self.device is not None:
            self.torch.cuda.synchronize(self.device)
        self.dist.barrier()

    def max_float(self, x):
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allgather_bytes(self, payload):
        """payload: bytes -> list of bytes from every rank (sizes may differ)."""
        torch, dist = self.torch, self.dist
        n = torch.tensor([len(payload)], dtype=torch.int64, device=self._dev())
        sizes = [torch.zeros(1, dtype=torch.int64, device=self._dev()) for _ in range(self.world)]
        dist.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
        mx = max(max(sizes), 1)
        buf = torch.zeros(mx, dtype=torch.uint8)
        if len(payload):
            buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        buf = buf.to(self._dev())
        out = [torch.empty(mx, dtype=torch.uint8, device=self._dev()) for _ in range(self.world)]
        dist.all_gather(out, buf)
        return [bytes(o[:s].cpu().numpy().tobytes()) for o, s in zip(out, sizes)]

    def allgather_device(self, tensor):
        """tensor: uint8 (device tensor with nccl, CPU tensor with gloo), sizes may differ between ranks.
        Returns (list of uint8 tensors, list of byte counts); the tensors are padded to the largest size."""
        torch, dist = self.torch, self.dist
        if self.world == 1:
            return [tensor], [tensor.numel()]
        # sizes: one collective, one host read
        n = torch.tensor([tensor.numel()], dtype=torch.int64, device=self._dev())
        sizes_t = torch.empty(self.world, dtype=torch.int64, device=self._dev())
        dist.all_gather_into_tensor(sizes_t, n)
        sizes = [int(x) for x in sizes_t.tolist()]
        mx = max(max(sizes), 1)
        if tensor.numel() == mx:
            buf = tensor
        else:
            buf = torch.zeros(mx, dtype=torch.uint8, device=self._dev())
            buf[:tensor.numel()] = tensor
        # one receive buffer, kept between calls while it is large enough (a fresh allocation per step would cost
        # more than the collective at these sizes)
        need = mx * self.world
        if getattr(self, "_gather_buf", None) is None or self._gather_buf.numel() < need or self._gather_buf.device != buf.device:
            self._gather_buf = torch.empty(need + need // 4, dtype=torch.uint8, device=self._dev())
        flat = self._gather_buf[:need]
        dist.all_gather_into_tensor(flat, buf)
        if self.device is not None:
            torch.cuda.synchronize(self.device)
        return [flat[r * mx:(r + 1) * mx] for r in range(self.world)], sizes

    def ring_shift(self, send, recv_like):
        """rank r -> r+1 (no wrap).  `send`: uint8 tensor or None (last rank); returns the received uint8
        tensor or None (rank 0).  Fixed size on every rank."""
        dist = self.dist
        ops = []
        recv = None
        if self.rank + 1 < self.world and send is not None:
            ops.append(dist.P2POp(dist.isend, send, self.rank + 1))
        if self.rank > 0:
            recv = self.torch.empty_like(recv_like)
            ops.append(dist.P2POp(dist.irecv, recv, self.rank - 1))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if self.device is not None:
            self.torch.cuda.synchronize(self.device)
        return recv

    def allreduce_min_max(self, t_min, t_max):
        self.dist.all_reduce(t_min, op=self.dist.ReduceOp.MIN)
        self.dist.all_reduce(t_max, op=self.dist.ReduceOp.MAX)
        if self.device is not None:
            self.torch.cuda.synchronize(self.device)

    def device_bytes(self, ptr, nbytes):
        """uint8 torch tensor aliasing raw device memory [ptr, ptr+nbytes)."""
        return self.torch.as_tensor(_DevArray(ptr, nbytes), device=self._dev())

    def device_i32(self, ptr, n):
        return self.device_bytes(ptr, 4 * n).view(self.torch.int32)


# ------------------------------------------------------------------------------------------------
# engine over libcontrack_hip.so
# ------------------------------------------------------------------------------------------------
class HipShardEngine:
    """The per-shard stages of include/contrack_hip.h on one GPU; buffers stay in HBM.  With a CPU
    communicator (gloo; used to test the sharded HIP stages with several processes on ONE GPU) the small
    exchanged buffers are staged through host memory."""

    def __init__(self, tracker, comm, anom_dev, T, ny, nx, thr, cmp_op, wrow, flag_dev):
        self.trk, self.comm = tracker, comm
        self.anom_dev, self.flag_dev = anom_dev, flag_dev
        self.T, self.ny, self.nx = int(T), int(ny), int(nx)
        self.thr, self.cmp_op, self.wrow = thr, cmp_op, wrow
        self.on_device = comm.device is not None
        self._keep = None

    def label2d(self, has_prev):
        self.trk.shard_label2d(self.anom_dev, self.T, self.ny, self.nx, self.thr, self.cmp_op, self.wrow, has_prev)

    def halo_nbytes(self):
        return self.trk.halo_size()

    def halo_export(self):
        ptr, _ = self.trk.halo_export()
        if self.on_device:
            return self.comm.device_bytes(ptr.value, self.halo_nbytes())
        host = np.empty(self.halo_nbytes(), dtype=np.uint8)
        self.trk.d2h(host, ptr)
        return self.comm.torch.from_numpy(host)

    def halo_template(self):
        return self.comm.torch.empty(self.halo_nbytes(), dtype=self.comm.torch.uint8, device=self.comm._dev())

    def halo_import(self, tensor):
        if self.on_device:
            self.trk.halo_import(C.c_void_p(tensor.data_ptr()), tensor.numel())
        else:
            stage = self.trk.malloc(tensor.numel())
            try:
                self.trk.h2d(stage, tensor.numpy())
                self.trk.halo_import(stage, tensor.numel())
            finally:
                self.trk.free(stage)

    def overlap(self):
        self.trk.shard_overlap()

    def tables(self):
        return self.trk.shard_tables()

    def resolve_gathered(self, shard, t_begin, overlap, twosided):
        """device path: table blob stays in HBM, all-gathered over the communicator, resolved on the device"""
        ptr, nbytes = self.trk.shard_tables_dev()
        if self.on_device:
            mine = self.comm.device_bytes(ptr.value, nbytes)
        else:
            host = np.empty(nbytes, dtype=np.uint8)
            self.trk.d2h(host, ptr)
            mine = self.comm.torch.from_numpy(host)
        blobs, sizes = self.comm.allgather_device(mine)
        self._gathered = blobs                                   # keep alive until the resolver is done
        stage = []
        if self.on_device:
            ptrs = [b.data_ptr() for b in blobs]
        else:
            ptrs = []
            for b, n in zip(blobs, sizes):
                d = self.trk.malloc(max(n, 8))
                self.trk.h2d(d, b.numpy()[:n])
                stage.append(d)
                ptrs.append(d.value)
        try:
            ext, n = self.trk.shard_resolve_dev(ptrs, sizes, shard, t_begin, overlap, twosided)
        finally:
            for d in stage:
                self.trk.free(d)
        self.trk.sync()
        return self._wrap_ext(ext, n)

    def _wrap_ext(self, ptr, n):
        self._ext = (ptr, n)
        if self.on_device:
            ext = self.comm.device_i32(ptr.value, 2 * (n + 1))
        else:
            host = np.empty(2 * (n + 1), dtype=np.int32)
            self.trk.d2h(host, ptr)
            ext = self.comm.torch.from_numpy(host)
            self._keep = ext
        return ext[:n + 1], ext[n + 1:]

    def stats(self):
        return self.trk.stats()

    def extents(self, result, shard, t_begin):
        ptr, n = self.trk.shard_extents(result, shard, t_begin)
        self.trk.sync()
        self._ext = (ptr, n)
        if self.on_device:
            ext = self.comm.device_i32(ptr.value, 2 * (n + 1))
        else:
            host = np.empty(2 * (n + 1), dtype=np.int32)
            self.trk.d2h(host, ptr)
            ext = self.comm.torch.from_numpy(host)
            self._keep = ext
        return ext[:n + 1], ext[n + 1:]

    def write(self, persistence):
        if not self.on_device and self._keep is not None:
            self.trk.h2d(self._ext[0], self._keep.numpy())          # the all-reduced extents
        return self.trk.shard_write(persistence, self.flag_dev)


# ------------------------------------------------------------------------------------------------
# the driver
# ------------------------------------------------------------------------------------------------
def run_sharded(engine, comm, t_begin, overlap, persistence, twosided, device_resolve=True):
    """Runs the whole path for this rank's shard.  Returns (n_tracked, info) -- identical on all ranks."""
    rank, world = comm.rank, comm.world
    engine.label2d(has_prev=rank > 0)
    if world > 1:
        send = engine.halo_export() if rank + 1 < world else None
        recv = comm.ring_shift(send, engine.halo_template())
        if rank > 0:
            engine.halo_import(recv)
    engine.overlap()
    done = False
    if device_resolve and hasattr(engine, "resolve_gathered"):
        # tables stay in HBM: all-gather of the device blobs, resolver replicated on every GPU
        try:
            tmin, tmax = engine.resolve_gathered(rank, t_begin, overlap, twosided)
            info = engine.stats()
            done = True
        except ValueError:
            # the device resolver gave up (removal cascade longer than its pass budget).  The condition is a
            # function of the gathered tables, identical on every rank: all ranks fall back together.
            done = False
    if not done:
        blob = engine.tables()
        blobs = comm.allgather_bytes(blob) if world > 1 else [blob]
        result = _native.resolve(blobs, overlap, twosided)
        tmin, tmax = engine.extents(result, rank, t_begin)
        info = result.info()
        result.free()
    if world > 1:
        comm.allreduce_min_max(tmin, tmax)
    n_alive, wrote_bg = engine.write(persistence)
    bg = comm.max_float(1.0 if wrote_bg else 0.0) > 0 if world > 1 else wrote_bg
    return n_alive + (1 if bg else 0) - 1, info            # len(np.unique(flag)) - 1, contrack.py:793


# ------------------------------------------------------------------------------------------------
# bench.py leg for N > 1 (launched by torch.distributed.run, one rank per GPU)
# ------------------------------------------------------------------------------------------------
def bench_main(args, wl, workloads, hbm_peak):
    import torch
    import torch.distributed as dist
    from . import synth
    # Keep stdout clean for the ONE JSON line: RCCL prints a version banner through C stdio at communicator
    # creation.  Everything written to fd 1 until the result is ready goes to stderr instead.
    sys_stdout_fd = os.dup(1)
    os.dup2(2, 1)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    local = int(os.environ.get("LOCAL_RANK", rank))
    backend = os.environ.get("CTK_DIST_BACKEND", "nccl")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        comm = TorchComm(device=local)
    else:
        # plumbing check on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device): all ranks share
        # GPU 0, the small exchanged buffers travel through gloo on the host.  Timings are meaningless then.
        local = 0
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
        comm = TorchComm(device=None)
    T, ny, nx = wl["T"], wl["ny"], wl["nx"]
    weak = getattr(args, "scaling", "weak") == "weak"
    if weak:
        # weak scaling: one member of wl["T"] steps per GPU, members concatenated on the time axis (the layout of
        # BASELINE.json configs[4]); rank r holds member r = synthetic slab with seed r, global steps [r T, (r+1) T).
        # The halo exchange, the table all-gather and the resolve run over the whole N x T axis.
        t0, t1 = rank * T, (rank + 1) * T
        a = synth.smooth_field(T, ny, nx, seed=rank)
        T_total = T * world
    else:
        t0, t1 = shard_bounds(T, world)[rank]
        a = synth.smooth_field(T, ny, nx, seed=0)[t0:t1]
        T_total = T
    lat, _ = synth.grid(ny, nx)
    w = np.array((111 * np.float32(180.0 / (ny - 1)) * 111 * np.float32(360.0 / nx) * np.cos(lat * np.pi / 180))).astype(np.float32)
    thr = np.full(t1 - t0, np.float64(np.float32(wl["threshold"])))
    trk = _native.Tracker(local)
    d_in = trk.malloc(max(a.nbytes, 8))
    d_out = trk.malloc(max(a.nbytes, 8))
    trk.h2d(d_in, a)
    trk.set_timing(1)            # HIP events around the two streaming kernels only (see bench.py)
    eng = HipShardEngine(trk, comm, d_in, t1 - t0, ny, nx, thr, _native.CMP_OPS[wl["gorl"]], w, d_out)

    def step():
        return run_sharded(eng, comm, t0, wl["overlap"], wl["persistence"], wl["twosided"])

    # communicator warm-up (not a step): collectives and point-to-point channels are created lazily by RCCL
    comm.barrier()
    comm.allgather_bytes(b"x")
    comm.ring_shift(torch.zeros(8, dtype=torch.uint8, device=comm._dev()) if rank + 1 < world else None,
                    torch.zeros(8, dtype=torch.uint8, device=comm._dev()))
    for _ in range(args.warmup):
        n_tracked, info = step()
    if args.warmup == 0:
        n_tracked, info = None, None
    comm.barrier()
    trk.sync()
    acc = {}
    tb = time.perf_counter()
    for _ in range(args.steps):
        n_tracked, info = step()
        for k, v in trk.timings().items():
            acc[k] = acc.get(k, 0.0) + v
    trk.sync()
    comm.barrier()
    dt = comm.max_float(time.perf_counter() - tb)
    per = {k: v / args.steps for k, v in acc.items()}
    px = (t1 - t0) * ny * nx
    alg = {"k_threshold": 4.0 * px, "k_relabel": 4.0 * px}
    kern = max(alg, key=lambda k: per.get(k, 0.0))
    achieved = alg[kern] / (per[kern] * 1e-3) / 1e9 if per.get(kern, 0) > 0 else 0.0
    if rank == 0:
        out = dict(metric="timesteps/sec labeled+tracked", value=T_total * args.steps / dt, unit="timesteps/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=dt * 1e3 / args.steps, higher_is_better=True,
                   scaling="weak" if weak else "strong", vs_baseline=None, dtype="f32 compare / int32 labels / int64 exact areas", data="synthetic",
                   config=dict(workload="%s: %s%dx%dx%d float32, threshold %s %g, overlap %g, persistence %d, twosided %s" % (
                       args.workload, ("%d members concatenated on the time axis, each " % world) if weak else "", T, ny, nx, wl["gorl"],
                       wl["threshold"], wl["overlap"], wl["persistence"], wl["twosided"]),
                       total_timesteps=T_total,
                       parallelism="time-sharded x%d (one-timestep halo + table all-gather over RCCL)" % world, n_tracked=n_tracked,
                       resolve_info=info),
                   roofline=dict(bound="hbm", kernel=kern, achieved=achieved, peak=hbm_peak, unit="GB/s", frac=achieved / hbm_peak,
                                 traffic=None, algorithmic_bytes_per_launch=alg[kern], avg_kernel_ms=per.get(kern),
                                 note="rank 0's shard"),
                   kernels_ms=per)
        try:
            C.CDLL(None).fflush(None)            # drain C stdio (RCCL banner) into stderr before stdout is restored
        except Exception:
            pass
        os.dup2(sys_stdout_fd, 1)
        print(json.dumps(out), flush=True)
    trk.free(d_in)
    trk.free(d_out)
    trk.close()
    dist.destroy_process_group()
