"""nccl plumbing of contrack_amd/dist.py: a torch tensor built on raw library memory must alias it.
Own module (own process order): torch.cuda has to be initialised BEFORE libcontrack_hip.so creates its
handle -- torch bundles its own ROCm runtime libraries, and whichever HIP runtime is loaded first serves
the process (contrack_amd.dist.bench_main follows the same order)."""
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_torch_tensor_aliases_library_memory():
    """the nccl path all-reduces the per-id extents IN PLACE through a torch view of the library's buffer"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible to torch")
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    from contrack_amd import _native, dist as cdist
    comm = types.SimpleNamespace(torch=torch, device=0, _dev=lambda: torch.device("cuda", 0))
    trk = _native.Tracker(0)
    p = trk.malloc(64 * 4)
    try:
        trk.h2d(p, np.arange(64, dtype=np.int32))
        t = cdist.TorchComm.device_bytes(comm, p.value, 256).view(torch.int32)
        assert t.is_cuda and t.data_ptr() == p.value
        assert torch.equal(t.cpu(), torch.arange(64, dtype=torch.int32))
        t.mul_(2)
        torch.cuda.synchronize()
        back = np.empty(64, dtype=np.int32)
        trk.d2h(back, p)
        assert np.array_equal(back, 2 * np.arange(64, dtype=np.int32))
    finally:
        trk.free(p)
        trk.close()
