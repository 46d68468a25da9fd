"""The STAGED C API (ctk_shard_label2d -> halo -> ctk_shard_overlap -> ctk_shard_tables -> ctk_resolve on the host ->
ctk_shard_extents -> ctk_shard_write) with random shard boundaries, driven in one process (tests/shard_inproc.py): the
table-level protocol that ctk_resolve specifies.  The product's multi-GPU path is ctk_track_sharded_* (tests/test_gpu_sharded.py);
the table-only host resolver cannot see pixels, so decisions on rounded area sums within rounding distance of the threshold
are REPORTED by it (n_ambiguous) -- the sharded product path resolves them."""
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
def test_random_shard_boundaries_staged_api(handles, oracle_lib, i):
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
    ties = ref.stats()["exact_fixups"] > 0
    got, ng, info = sharded(handles[:n], a, thrv, op, w, ov, pers, two, cuts)
    if ties:
        # decisions on rounded sums near the threshold: the table-level resolver reports the ones whose sums it had to round
        # (the product paths re-evaluate them); it may still come out equal
        assert info["n_ambiguous"] > 0 or (np.array_equal(got, want) and ng == nw)
    else:
        assert np.array_equal(got, want) and ng == nw, cuts
