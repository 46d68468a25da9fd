"""BASELINE.json configs[2] at its OWN size on one GPU: 14 600 x 721 x 1440 float32 = 1.516e10 pixels, 60.6 GB in + 60.6 GB out.
From 2^31 - 2 elements on the reference itself changes behaviour (scipy labels in int64, contrack.py:687, :751); here every
pixel offset is 64-bit and the ids stay int32 (they are counted, not addressed).  Three kinds of evidence past 2^31 / 2^32 / 2^33 pixels:

  (a) embedding: the slab is background except four windows that hold 0.25 deg cases the C oracle can run (66 / 50 steps), one
      before pixel 2^31, one beyond 2^32, one beyond 2^33, one at the very end -- `flag` inside every window must equal the oracle's (ids included: scipy
      numbers the components over the whole slab, so a window's ids are the oracle's plus the labels of the windows in front of
      it) and be zero everywhere else;
  (b) the device-generated configs[2] slab: one call against four time shards (the product's ctk_track_sharded_*), equal 64-bit
      position-weighted checksums per shard window and equal n_tracked;
  (c) size-independent properties of that result, evaluated on the device: flag is a subset of the mask (float64 compare,
      contrack.py:665), ids in range, number of distinct ids = n_tracked, no id with a time extent below `persistence`
      (contrack.py:765-772).
  (d) the host-array entry on a 2.18e9-element host slab (both result transfers) against the device-resident entry.
Needs ~150 GB of the 288 GB of HBM and ~30 GB of host memory."""
import ctypes as C
import threading

import numpy as np
import pytest

from contrack_amd import _native, synth
from contrack_amd.dist import shard_bounds

pytestmark = pytest.mark.gpu

T, NY, NX = 14600, 721, 1440
PLANE = NY * NX
PERSISTENCE = 20


def _off(p, nbytes):
    return C.c_void_p(p.value + int(nbytes))


@pytest.fixture(scope="module")
def big():
    """one handle + the two 60.6 GB device slabs, shared by the tests of this module"""
    if _native.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the GPU box")
    trk = _native.Tracker(0)
    d_in = d_out = None
    try:
        d_in = trk.malloc(T * PLANE * 4)
        d_out = trk.malloc(T * PLANE * 4)
    except MemoryError as e:
        if d_in is not None:
            trk.free(d_in)
        trk.close()
        pytest.fail("the configs[2] slab needs 2 x 60.6 GB of device memory: %s" % e)
    lat, _ = synth.grid(NY, NX)
    from contrack_amd.contrack import row_weights
    w = row_weights(lat, np.float32(0.25), np.float32(0.25))
    yield trk, d_in, d_out, w
    trk.free(d_in)
    trk.free(d_out)
    trk.close()


def test_embedded_windows_beyond_2p31_2p32_2p33_match_oracle(big, oracle_lib):
    trk, d_in, d_out, w = big
    thr_val = 160.0
    # the cases: (steps, seed); every window = one background step + the case + one background step, so that the oracle filters the
    # case's first and last step exactly as the big slab does (contrack.py:710: steps 1 .. T-2 are filtered)
    cases = {"A": (64, 5), "B": (48, 6)}
    want, nlab = {}, {}
    small = _native.Tracker(0)
    try:
        for k, (n, seed) in cases.items():
            a = np.full((n + 2, NY, NX), -1000.0, dtype=np.float32)
            a[1:-1] = synth.smooth_field(n, NY, NX, seed=seed)
            thr = oracle_lib.prepare_thresholds(thr_val, n + 2)
            f, nt = oracle_lib.run_contrack(a, thr, ">=", w, 0.5, PERSISTENCE, True)
            g, ng = small.track(a, thr, 0, w, 0.5, PERSISTENCE, True)           # (also gives the number of 3-D labels before persistence)
            assert np.array_equal(g, f) and ng == nt and nt > 0
            want[k] = (a, f, nt)
            nlab[k] = small.stats()["labels_3d"]
    finally:
        small.close()
    # windows: (first step, case).  2^31 px = step 2068.4, 2^32 px = step 4136.7, 2^33 px = step 8273.4
    windows = [(100, "A"), (4300, "B"), (8400, "A"), (T - 66 - 2, "A")]
    assert 4300 * PLANE > 2 ** 32 and 8400 * PLANE > 2 ** 33 and (100 + 66) * PLANE < 2 ** 31
    trk.memset(d_in, 0, T * PLANE * 4)                                           # 0.0 < 160: background
    trk.memset(d_out, 0xff, T * PLANE * 4)                                        # (every pixel of the result must be WRITTEN)
    for t0, k in windows:
        trk.h2d(_off(d_in, t0 * PLANE * 4), want[k][0])
    thr = oracle_lib.prepare_thresholds(thr_val, T)
    n = trk.track_dev(d_in, T, NY, NX, thr, 0, w, 0.5, PERSISTENCE, True, d_out)
    st = trk.stats()
    assert n == sum(want[k][2] for _, k in windows)
    assert st["labels_3d"] == sum(nlab[k] for _, k in windows)
    base, nz_expected = 0, 0
    for t0, k in windows:
        a, f, nt = want[k]
        got = np.empty(f.shape, dtype=np.int32)
        trk.d2h(got, _off(d_out, t0 * PLANE * 4))
        expect = np.where(f > 0, f + np.int32(base), 0).astype(np.int32)          # scipy numbers over the whole slab
        assert np.array_equal(got, expect), ("window at step %d" % t0)
        base += nlab[k]
        nz_expected += int(np.count_nonzero(f))
    # zero everywhere else: the nonzero count of the whole result is the windows'
    _, nz = trk.checksum_i32(d_out, T * PLANE, 0)
    assert nz == nz_expected
    # ... and the device-side properties hold
    pr = trk.check_flag(d_in, d_out, T, NY, NX, thr, 0, PERSISTENCE, st["labels_3d"])
    assert pr["flag_outside_mask"] == 0 and pr["ids_out_of_range"] == 0 and pr["ids_below_persistence"] == 0
    assert pr["ids"] == n and pr["nonzero"] == nz_expected


def test_configs2_slab_one_call_vs_four_shards_and_properties(big):
    trk, d_in, d_out, w = big
    trk.synth_fill(d_in, T, NY, NX, seed=0)
    thr = np.full(T, np.float64(np.float32(160.0)))
    trk.memset(d_out, 0xff, T * PLANE * 4)
    n_one = trk.track_dev(d_in, T, NY, NX, thr, 0, w, 0.5, PERSISTENCE, True, d_out)
    st = trk.stats()
    assert n_one > 100, st
    # (c) properties of the one-call result
    pr = trk.check_flag(d_in, d_out, T, NY, NX, thr, 0, PERSISTENCE, st["labels_3d"])
    assert pr["flag_outside_mask"] == 0 and pr["ids_out_of_range"] == 0, pr
    assert pr["ids_below_persistence"] == 0 and pr["ids"] == n_one, (pr, n_one)
    assert 0.02 < pr["nonzero"] / float(T * PLANE) < 0.2, pr                    # a real workload: a few per cent of the pixels are tracked
    # (b) four time shards (threads = ranks, in-process communicator) write into the same output buffer after it was poisoned again
    world = 4
    bounds = shard_bounds(T, world)
    ref = [trk.checksum_i32(_off(d_out, t0 * PLANE * 4), (t1 - t0) * PLANE, t0 * PLANE) for t0, t1 in bounds]
    trk.memset(d_out, 0xff, T * PLANE * 4)
    hs = [_native.Tracker(0) for _ in range(world)]
    group = _native.CommGroup(world)
    comms = [_native.Comm.local(hs[r], group, r) for r in range(world)]
    res, err = [None] * world, [None] * world

    def work(r):
        t0, t1 = bounds[r]
        try:
            res[r] = hs[r].track_sharded_dev(comms[r], _off(d_in, t0 * PLANE * 4), t1 - t0, t0, T, NY, NX, thr[t0:t1].copy(), 0, w, 0.5, PERSISTENCE, True,
                                             _off(d_out, t0 * PLANE * 4))
        except Exception as e:                                                   # noqa: BLE001 -- reported below
            err[r] = e

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for c in comms:
        c.close()
    group.close()
    for h in hs:
        h.close()
    for e in err:
        if e is not None:
            raise e
    assert res == [n_one] * world, (res, n_one)
    got = [trk.checksum_i32(_off(d_out, t0 * PLANE * 4), (t1 - t0) * PLANE, t0 * PLANE) for t0, t1 in bounds]
    assert got == ref


def test_host_array_entry_past_2p31_elements(big):
    """ctk_track_f32 on a host slab of 2.18e9 elements (2100 x 721 x 1440): the result as run tables expanded by host threads, and the
    dense copy, against the device-resident entry on the same slab (position-weighted checksums of the whole result)."""
    trk, d_in, d_out, w = big
    T2 = 2100
    n = T2 * PLANE
    assert n > 2 ** 31
    trk.synth_fill(d_in, T2, NY, NX, seed=5)
    thr = np.full(T2, np.float64(np.float32(160.0)))
    trk.memset(d_out, 0xff, n * 4)
    n_dev = trk.track_dev(d_in, T2, NY, NX, thr, 0, w, 0.5, PERSISTENCE, True, d_out)
    ref = trk.checksum_i32(d_out, n, 0)
    assert n_dev > 10 and ref[1] > 0
    a = np.empty((T2, NY, NX), dtype=np.float32)
    trk.d2h(a, d_in)
    try:
        for mode in (1, 0):
            trk.set_result_transfer(mode)
            flag, n_host = trk.track(a, thr, 0, w, 0.5, PERSISTENCE, True)
            assert (trk.stats()["result_as_runs"] >= 1) == (mode == 1)
            assert n_host == n_dev
            trk.memset(d_out, 0xff, n * 4)
            trk.h2d(d_out, flag)
            assert trk.checksum_i32(d_out, n, 0) == ref, mode
            del flag
    finally:
        trk.set_result_transfer(-1)
        trk.release_io()
