"""Time-sharded stages of the C API with random shard boundaries (shards of a single step included), device and host
resolver, against the one-call result -- all in one process (tests/shard_inproc.py)."""
import numpy as np
import pytest

from contrack_amd import _native
from shard_inproc import sharded
from test_gpu_parity import _edge_case, _random_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handles():
    hs = [_native.Tracker(0) for _ in range(7)]
    yield hs
    for h in hs:
        h.close()


@pytest.mark.parametrize("i", range(48))
def test_random_shard_boundaries(handles, oracle_lib, i):
    a, thr, gorl, ov, pers, two = _random_case(i) if i % 3 else _edge_case(i)
    T, ny, nx = a.shape
    if T < 2:
        pytest.skip("single step")
    rng = np.random.default_rng(7000 + i)
    n = int(rng.integers(2, min(6, T) + 1))
    cuts = [0] + [int(v) for v in np.sort(rng.choice(np.arange(1, T), size=n - 1, replace=False))] + [T]
    lat = np.linspace(90, -90, ny).astype(np.float32)
    w = oracle_lib.row_weights(lat, np.float32(180.0 / max(ny - 1, 1)), np.float32(360.0 / nx))
    thrv = oracle_lib.prepare_thresholds(thr, T)
    op = _native.CMP_OPS[gorl]
    ref = handles[6]
    want, nw = ref.track(a, thrv, op, w, ov, pers, two)
    if ref.stats()["exact_fixups"]:
        pytest.skip("exact ties on pole-touching components: re-evaluated on one GPU only (DESIGN.md, exact areas)")
    for dev in (True, False):
        got, ng = sharded(handles[:n], a, thrv, op, w, ov, pers, two, cuts, dev)
        assert np.array_equal(got, want) and ng == nw, (cuts, dev)
