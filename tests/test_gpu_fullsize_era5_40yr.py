"""BASELINE.json configs[3] at its OWN size on one GPU: ERA5 0.25 deg, 6-hourly, 40 years = 58 400 x 721 x 1440 float32
= 6.06e10 pixels (beyond 2^35), 242 GB in + 242 GB out.  The slab does not fit HBM twice: it passes through the streaming entry
(ctk_track_stream_cb, next row N4) -- the reader hands it over chunk by chunk, only the bit mask (7.8 GB) and the run / component
tables stay resident, the flags leave chunk by chunk through the writer.

Embedding: the slab is background except four windows holding 0.25 deg cases the C oracle can run: one near the start, one beyond
pixel 2^34 (step 16 548), one beyond 2^35 (step 33 096), one at the very end.  The writer checks every chunk: inside a window the
flags equal the oracle's (ids offset by the labels of the windows in front), everywhere else they are zero.

Host side: the reader fills the library's pinned chunk with the background once per buffer (it remembers which buffers already are
background and only repairs the rows a window dirtied), so the test is bound by PCIe (2 x 242 GB at ~55 GB/s), not by host fills."""
import numpy as np
import pytest

from contrack_amd import _native, synth
from contrack_amd.contrack import row_weights

pytestmark = pytest.mark.gpu

T, NY, NX = 58400, 721, 1440
PLANE = NY * NX
PERSISTENCE = 20


def test_configs3_slab_through_the_streaming_entry(oracle_lib):
    if _native.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the GPU box")
    assert T * PLANE > 2 ** 35
    lat, _ = synth.grid(NY, NX)
    w = row_weights(lat, np.float32(0.25), np.float32(0.25))
    thr_val = 160.0
    cases = {"A": (64, 5), "B": (48, 6)}
    want, nlab = {}, {}
    with _native.Tracker(0) as small:
        for k, (n, seed) in cases.items():
            a = np.full((n + 2, NY, NX), -1000.0, dtype=np.float32)
            a[1:-1] = synth.smooth_field(n, NY, NX, seed=seed)
            thr = oracle_lib.prepare_thresholds(thr_val, n + 2)
            f, nt = oracle_lib.run_contrack(a, thr, ">=", w, 0.5, PERSISTENCE, True)
            g, ng = small.track(a, thr, 0, w, 0.5, PERSISTENCE, True)
            assert np.array_equal(g, f) and ng == nt and nt > 0
            want[k] = (a, f, nt)
            nlab[k] = small.stats()["labels_3d"]
    windows = [(100, "A"), (17000, "B"), (34000, "A"), (T - 66, "A")]
    assert 17000 * PLANE > 2 ** 34 and 34000 * PLANE > 2 ** 35
    base, expect = 0, []
    for t0, k in windows:
        a, f, nt = want[k]
        expect.append((t0, t0 + f.shape[0], a, np.where(f > 0, f + np.int32(base), 0).astype(np.int32)))
        base += nlab[k]

    clean = {}                         # address of a pinned input chunk -> list of (first, last) local steps that are NOT background
    stats = {"reads": 0, "writes": 0, "nonzero": 0, "bad": []}

    def reader(t0, nt, dst):
        stats["reads"] += 1
        key = dst.ctypes.data
        dirty = clean.get(key)
        if dirty is None or dirty == "all" or dst.shape[0] > clean.get((key, "n"), 0):
            dst[...] = 0.0                                           # 0.0 < 160: background
        else:
            for a0, a1 in dirty:
                dst[a0:a1] = 0.0
        now = []
        for w0, w1, a, _ in expect:
            lo, hi = max(w0, t0), min(w1, t0 + nt)
            if lo < hi:
                dst[lo - t0:hi - t0] = a[lo - w0:hi - w0]
                now.append((lo - t0, hi - t0))
        clean[key] = now
        clean[(key, "n")] = max(dst.shape[0], clean.get((key, "n"), 0))

    def writer(t0, nt, flags):
        stats["writes"] += 1
        nz = int(np.count_nonzero(flags))
        inside = 0
        for w0, w1, _, f in expect:
            lo, hi = max(w0, t0), min(w1, t0 + nt)
            if lo < hi:
                if not np.array_equal(flags[lo - t0:hi - t0], f[lo - w0:hi - w0]):
                    stats["bad"].append((w0, lo, hi))
                inside += int(np.count_nonzero(f[lo - w0:hi - w0]))
        if nz != inside:
            stats["bad"].append(("nonzero outside the windows", t0, nt, nz, inside))
        stats["nonzero"] += nz

    thr = oracle_lib.prepare_thresholds(thr_val, T)
    with _native.Tracker(0) as trk:
        _, n = trk.track_stream(reader, thr, 0, w, 0.5, PERSISTENCE, True, sink=writer, shape=(T, NY, NX), dtype=np.float32)
        st = trk.stats()
        ms = trk.stream_times()
    assert not stats["bad"], stats["bad"][:4]
    assert n == sum(want[k][2] for _, k in windows)
    assert st["labels_3d"] == sum(nlab[k] for _, k in windows)
    assert stats["nonzero"] == sum(int(np.count_nonzero(f)) for _, _, _, f in expect)
    assert stats["reads"] == stats["writes"] and stats["reads"] >= T * PLANE * 4 // (300 << 20)
    print("configs[3] streamed: %d chunks, input phase %.1f s, output phase %.1f s" % (stats["reads"], ms["input_phase"] / 1e3, ms["output_phase"] / 1e3))
