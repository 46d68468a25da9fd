"""Streaming entries (next row N4, ctk_track_stream_*): the slab passes through chunk-sized device buffers, only mask and
tables stay resident; results must equal the one-call path bit for bit, whatever the chunking."""
import numpy as np
import pytest

import golden_util
from contrack_amd import _native, synth
from contrack_amd.contrack import row_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def trk():
    with _native.Tracker(0) as t:
        yield t


@pytest.mark.parametrize("name", ["refslab_two", "syn2deg_le", "thr_vector", "noise", "chain_a", "odd_17x64", "f64pole_blocky"])
@pytest.mark.parametrize("chunk", [1, 3, 0])
def test_stream_arrays_equal_golden(trk, name, chunk):
    g = golden_util.load(name)
    f64 = g["anom"].dtype == np.float64
    flag, n = trk.track_stream(g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"],
                               chunk_steps=chunk)
    assert np.array_equal(flag, g["flag"]) and n == len(np.unique(g["flag"])) - 1, (name, chunk, f64)


@pytest.mark.parametrize("chunk", [2, 5])
def test_stream_callbacks(trk, chunk):
    """reader / writer callbacks (what a netCDF variable read and written slice by slice looks like to the library)"""
    g = golden_util.load("refslab_two")
    T, ny, nx = g["anom"].shape
    reads, writes = [], []
    out = np.full((T, ny, nx), -7, dtype=np.int32)

    def reader(t0, nt, dst):
        reads.append((t0, nt))
        dst[...] = g["anom"][t0:t0 + nt]

    def writer(t0, nt, flags):
        writes.append((t0, nt))
        out[t0:t0 + nt] = flags
    _, n = trk.track_stream(reader, g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"], sink=writer,
                            shape=(T, ny, nx), dtype=g["anom"].dtype, chunk_steps=chunk)
    assert np.array_equal(out, g["flag"]) and n == len(np.unique(g["flag"])) - 1
    want = [(t0, min(chunk, T - t0)) for t0 in range(0, T, chunk)]
    assert reads == want and writes == want
    ms = trk.stream_times()
    assert ms["input_phase"] > 0 and ms["output_phase"] > 0


def test_stream_memmap_source_and_sink(trk, tmp_path):
    """files as source and sink: np.memmap stands in for the netCDF variables (no netCDF library in this image)"""
    T, ny, nx = 64, 181, 360
    lat, lon = synth.grid(ny, nx)
    anom = synth.smooth_field(T, ny, nx, seed=5)
    src = np.memmap(tmp_path / "anom.bin", dtype=np.float32, mode="w+", shape=(T, ny, nx))
    src[...] = anom
    src.flush()
    dst = np.memmap(tmp_path / "flag.bin", dtype=np.int32, mode="w+", shape=(T, ny, nx))
    wrow = row_weights(lat, 1.0, 1.0)
    thr = np.full(T, 160.0)
    want, nwant = trk.track(anom, thr, 0, wrow, 0.5, 3, True)
    got, n = trk.track_stream(np.memmap(tmp_path / "anom.bin", dtype=np.float32, mode="r", shape=(T, ny, nx)), thr, 0, wrow, 0.5, 3, True,
                              sink=dst, chunk_steps=7)
    dst.flush()
    assert n == nwant and np.array_equal(np.fromfile(tmp_path / "flag.bin", dtype=np.int32).reshape(T, ny, nx), want)


def test_stream_reader_failure_is_reported(trk):
    g = golden_util.load("refslab_two")

    def reader(t0, nt, dst):
        raise OSError("disk on fire")
    with pytest.raises(OSError, match="disk on fire"):
        trk.track_stream(reader, g["thr"], 0, g["wrow"], 0.5, 2, True, shape=g["anom"].shape, dtype=np.float32, chunk_steps=4)
    # the handle is usable afterwards
    flag, n = trk.track(g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
    assert np.array_equal(flag, g["flag"])


def test_stream_exact_fixups_reread_the_input(trk):
    """decisions on rounding boundaries re-stream the input once (documented): the reader sees every chunk twice"""
    g = golden_util.load("f64pole_blocky")
    T, ny, nx = g["anom"].shape
    reads = []
    out = np.zeros((T, ny, nx), dtype=np.int32)

    def reader(t0, nt, dst):
        reads.append(t0)
        dst[...] = g["anom"][t0:t0 + nt]
    _, n = trk.track_stream(reader, g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"], sink=out,
                            shape=(T, ny, nx), dtype=g["anom"].dtype, chunk_steps=4)
    assert np.array_equal(out, g["flag"]) and n == len(np.unique(g["flag"])) - 1
    st = trk.stats()
    assert st["exact_fixups"] > 0 and reads == 2 * list(range(0, T, 4))


@pytest.mark.parametrize("mode", [1, 0], ids=["runs", "dense"])
def test_result_memory_is_recycled_and_results_stay_valid(mode):
    """the binding recycles the memory of dropped results (with the dense copy: registered with HIP on reuse, one DMA instead of
    bounce buffers behind page faults; with the run transfer: touched pages for the host threads that write it); arrays still
    held -- and views of them -- are never overwritten"""
    import gc
    from contrack_amd import synth
    from contrack_amd.contrack import row_weights
    T, ny, nx = 40, 181, 360                                  # 10.4 MB of flags: above the pool's threshold
    a = synth.smooth_field(T, ny, nx, seed=3)
    b = synth.smooth_field(T, ny, nx, seed=4)
    lat, _ = synth.grid(ny, nx)
    w = row_weights(lat, np.float32(1.0), np.float32(1.0))
    thr = np.full(T, 160.0)
    with _native.Tracker(0) as t:
        t.set_result_transfer(mode)
        assert t.result_as_runs == (mode == 1)
        fa, na = t.track(a, thr, 0, w, 0.5, 3, True)
        keep_a = fa.copy()
        view = fa[5:7]                                           # a view keeps the block leased
        addr = fa.ctypes.data
        del fa
        gc.collect()
        fb, nb = t.track(b, thr, 0, w, 0.5, 3, True)             # must NOT land in the block `view` still uses
        assert fb.ctypes.data != addr and np.array_equal(view, keep_a[5:7])
        keep_b = fb.copy()
        del view, fb
        gc.collect()
        assert t._pool.hits == 0
        fa2, na2 = t.track(a, thr, 0, w, 0.5, 3, True)           # recycled block (registered now if the copy is dense)
        assert t._pool.hits == 1 and np.array_equal(fa2, keep_a) and na2 == na
        assert (t.stats()["result_as_runs"] >= 1) == (mode == 1)
        fb2, nb2 = t.track(b, thr, 0, w, 0.5, 3, True)
        assert t._pool.hits == 2 and np.array_equal(fb2, keep_b) and nb2 == nb and np.array_equal(fa2, keep_a)
        out = np.empty((T, ny, nx), np.int32)
        fo, _ = t.track(a, thr, 0, w, 0.5, 3, True, out=out)     # a caller-supplied array is used as it is
        assert fo is out and np.array_equal(out, keep_a)
    assert np.array_equal(fa2, keep_a) and np.array_equal(fb2, keep_b)       # still valid after the handle is gone
