"""Host-array entries: the result as run tables expanded on the host (default) against the dense slab written by k_relabel and
copied (ctk_set_result_transfer 0) -- the two must be the same array, and both the reference's (goldens).  The values are the
device's in both cases; only the transfer format differs."""
import numpy as np
import pytest

import golden_util
from contrack_amd import _native, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def trk():
    if _native.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the GPU box")
    t = _native.Tracker(0)
    yield t
    t.set_result_transfer(-1)
    t.close()


def _both(trk, anom, thr, op, wrow, overlap, persistence, twosided):
    trk.set_result_transfer(1)
    f1, n1 = trk.track(anom, thr, op, wrow, overlap, persistence, twosided)
    s1 = trk.stats()["result_as_runs"]
    trk.set_result_transfer(0)
    f0, n0 = trk.track(anom, thr, op, wrow, overlap, persistence, twosided)
    s0 = trk.stats()["result_as_runs"]
    trk.set_result_transfer(-1)
    assert s0 == 0 and s1 >= 1
    assert n0 == n1
    assert np.array_equal(f0, f1)
    return f1, n1, s1


@pytest.mark.parametrize("name", golden_util.case_names())
def test_runs_and_dense_copy_agree_on_the_goldens(trk, name):
    g = golden_util.load(name)
    f, n, _ = _both(trk, g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
    assert np.array_equal(f, g["flag"])
    assert n == len(np.unique(g["flag"])) - 1


def _field(T, ny, nx, seed, kind):
    if kind == "noise":
        return np.random.default_rng(seed).standard_normal((T, ny, nx)).astype(np.float32)
    return synth.smooth_field(T, ny, nx, seed=seed)


CASES = [
    # T, ny, nx, seed, kind, thr, overlap, persistence, twosided
    (400, 91, 180, 11, "smooth", 150.0, 0.5, 5, True),      # many blocks per lane
    (96, 181, 360, 12, "smooth", 160.0, 0.5, 5, True),
    (6, 721, 1440, 13, "smooth", 160.0, 0.5, 2, True),      # 23 words per row, fewer blocks than lanes
    (8, 181, 360, 14, "noise", 0.8, 0.5, 2, True),          # ~10^4 runs per timestep
    (1, 181, 360, 15, "smooth", 150.0, 0.5, 1, True),
    (40, 64, 128, 16, "smooth", 120.0, 0.3, 1, False),      # nx a multiple of 64: runs that end on a word boundary
    (30, 33, 67, 17, "smooth", 140.0, 0.5, 2, True),        # odd width (the generic write kernel on the dense side)
    (12, 16, 64, 18, "noise", 0.2, 0.5, 1, True),           # mostly foreground: full words, runs across words
    (3, 4, 4100, 19, "noise", 0.5, 0.5, 1, True),           # wider than 4096: the generic threshold / labelling kernels, 65 words per row
    (2, 700, 130, 20, "smooth", 120.0, 0.5, 1, False),      # tall and narrow
]


@pytest.mark.parametrize("case", CASES)
def test_runs_and_dense_copy_agree(trk, case):
    T, ny, nx, seed, kind, thr, overlap, persistence, twosided = case
    anom = _field(T, ny, nx, seed, kind)
    wrow = np.cos(np.deg2rad(np.linspace(-89, 89, ny))).astype(np.float32)
    f, n, _ = _both(trk, anom, np.full(T, thr), _native.CMP_OPS[">="], wrow, overlap, persistence, twosided)
    assert n == len(np.unique(f)) - 1


def test_blocks_with_complex_components_take_the_write_kernel(trk):
    """some golden of the seam-chain family must exercise the dense blocks (negative run values)"""
    seen = 0
    for name in golden_util.case_names():
        g = golden_util.load(name)
        trk.set_result_transfer(1)
        f, _ = trk.track(g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
        seen = max(seen, trk.stats()["result_as_runs"])
        assert np.array_equal(f, g["flag"])
    trk.set_result_transfer(-1)
    assert seen > 1


def test_all_foreground_reports_no_background(trk):
    """len(np.unique(flag)) - 1 counts the background only if a zero was written (contrack.py:793)"""
    T, ny, nx = 6, 16, 64
    anom = np.full((T, ny, nx), 5.0, dtype=np.float32)
    wrow = np.ones(ny, dtype=np.float32)
    f, n, _ = _both(trk, anom, np.full(T, 1.0), _native.CMP_OPS[">="], wrow, 0.5, 1, True)
    assert (f == 1).all() and n == 0


def test_f64_entry(trk):
    T, ny, nx = 50, 91, 180
    anom = synth.smooth_field(T, ny, nx, seed=21).astype(np.float64)
    wrow = np.cos(np.deg2rad(np.linspace(-89, 89, ny))).astype(np.float32)
    trk.set_result_transfer(1)
    f1, n1 = trk.track(anom, np.full(T, 150.0), _native.CMP_OPS[">="], wrow, 0.5, 3, True, f64=True)
    assert trk.stats()["result_as_runs"] >= 1
    trk.set_result_transfer(0)
    f0, n0 = trk.track(anom, np.full(T, 150.0), _native.CMP_OPS[">="], wrow, 0.5, 3, True, f64=True)
    trk.set_result_transfer(-1)
    assert n0 == n1 and np.array_equal(f0, f1) and n1 > 0


@pytest.mark.parametrize("chunk", [7, 64])
def test_streaming_entry_with_an_array_sink(trk, chunk):
    """ctk_track_stream_*: the slab arrives in chunks, the result leaves as run tables once the pass is over (a writer callback
    still gets dense chunks)"""
    T, ny, nx = 150, 91, 180
    anom = synth.smooth_field(T, ny, nx, seed=31)
    wrow = np.cos(np.deg2rad(np.linspace(-89, 89, ny))).astype(np.float32)
    thr = np.full(T, 150.0)
    want, nw = trk.track(anom, thr, 0, wrow, 0.5, 4, True)
    for mode in (1, 0):
        trk.set_result_transfer(mode)
        f, n = trk.track_stream(anom, thr, 0, wrow, 0.5, 4, True, chunk_steps=chunk)
        assert (trk.stats()["result_as_runs"] >= 1) == (mode == 1)
        assert n == nw and np.array_equal(f, want)
    trk.set_result_transfer(-1)


def test_unavailable_run_transfer_falls_back_to_the_dense_copy(trk):
    """a shortage of pinned memory for the lane buffers (or a failed lane copy) must not fail the call: the pass is repeated with the
    write kernel and the dense copy (mode 2 = the run transfer wanted but made unavailable; round-4 advisor finding)"""
    T, ny, nx = 120, 91, 180
    anom = synth.smooth_field(T, ny, nx, seed=77)
    wrow = np.cos(np.deg2rad(np.linspace(-89, 89, ny))).astype(np.float32)
    thr = np.full(T, 150.0)
    trk.set_result_transfer(1)
    want, nw = trk.track(anom, thr, 0, wrow, 0.5, 4, True)
    assert trk.stats()["result_as_runs"] >= 1 and nw > 0
    trk.set_result_transfer(2)
    f, n = trk.track(anom, thr, 0, wrow, 0.5, 4, True)
    st = trk.stats()
    assert st["result_as_runs"] == 0 and st["relabel_kernel"] >= 0
    assert n == nw and np.array_equal(f, want)
    f, n = trk.track_stream(anom, thr, 0, wrow, 0.5, 4, True, chunk_steps=16)
    assert trk.stats()["result_as_runs"] == 0 and n == nw and np.array_equal(f, want)
    trk.set_result_transfer(-1)
