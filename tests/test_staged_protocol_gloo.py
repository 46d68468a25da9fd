"""The STAGED, table-level protocol (tests/gloo_driver.py) under torch.distributed / gloo, world_size 2 and 3, on CPU: halo exchange,
table all-gather, the GPU-free host resolver of the library (ctk_resolve, csrc/ctk_resolve.cpp), extent all-reduce -- with a numpy
shard engine (tests/cpu_engine.py) standing in for the HIP stages.  What it covers: the table-level specification of the multi-rank
resolution and ctk_resolve itself.  What it does NOT cover: the product's N > 1 path, ctk_track_sharded_* (csrc/ctk_sharded.hip,
ctk_comm.hip) -- that needs a GPU and is tested in tests/test_gpu_sharded*.py (threads / processes on one GPU), tests/test_gpu_fullsize.py
and bench.py's in-run parity check; its GPU-free parts (boundary label resolution, rendezvous) in tests/test_shard_host.py.
RCCL with more than one rank has not run anywhere yet (one GPU per box here)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import golden_util


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from contrack_amd import dist as cdist
        import gloo_driver
        from cpu_engine import CpuShardEngine
        g = golden_util.load(name)
        T = g["anom"].shape[0]
        t0, t1 = cdist.shard_bounds(T, world)[rank]
        eng = CpuShardEngine(g["anom"][t0:t1], g["thr"][t0:t1], g["gorl"], g["wrow"])
        comm = gloo_driver.TorchComm(device=None)
        n, info = gloo_driver.run_sharded(eng, comm, t0, g["overlap"], g["persistence"], g["twosided"])
        ok = bool(np.array_equal(eng.flag, g["flag"][t0:t1])) and n == len(np.unique(g["flag"])) - 1
        q.put((rank, ok, n))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("syn2deg_s0", 2), ("busy_s0", 2), ("chain_a", 2), ("refslab_two", 3), ("T2", 3),
                                        ("noise_fwd", 2), ("all_fg", 2)])
def test_sharded_driver_gloo(name, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert len({n for _, _, n in res}) == 1


def test_shard_bounds():
    from contrack_amd import dist as cdist
    assert cdist.shard_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert cdist.shard_bounds(2, 3) == [(0, 1), (1, 2), (2, 2)]
    b = cdist.shard_bounds(2707, 8)
    assert b[0][0] == 0 and b[-1][1] == 2707 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
