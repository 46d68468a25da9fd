"""TEST INFRASTRUCTURE: the table-level sharded protocol of the staged C API (ctk_shard_tables -> ctk_resolve -> ctk_shard_extents ->
ctk_shard_write) driven over torch.distributed / gloo with world_size > 1 on CPU.  It exercises the GPU-free host resolver
on tables that come from several processes (halo exchange of the label map, all-gather of the table blobs, extent all-reduce).
The PRODUCT's multi-GPU path is ctk_track_sharded_* inside libcontrack_hip.so (contrack_amd/dist.py binds it; no torch there);
its GPU tests are tests/test_gpu_sharded.py."""
import numpy as np

from contrack_amd import _native


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



def run_sharded(engine, comm, t_begin, overlap, persistence, twosided):
    """Runs the whole path for this rank's shard.  Returns (n_tracked, info) -- identical on all ranks."""
    rank, world = comm.rank, comm.world
    engine.label2d(has_prev=rank > 0)
    if world > 1:
        send = engine.halo_export() if rank + 1 < world else None
        recv = comm.ring_shift(send, engine.halo_template())
        if rank > 0:
            engine.halo_import(recv)
    engine.overlap()
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


