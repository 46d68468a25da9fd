"""BASELINE.json configs[4] at its OWN size on one GPU: CESM-LE Z500, 40 members x 10 950 daily steps concatenated on the time axis
= 438 000 x 192 x 288 float32 = 2.42e10 pixels, 96.9 GB in + 96.9 GB out.  The grid has float64 irregular latitudes: the reference
refuses it unless set_up(force=True) was called (contrack.py:357-370), then dlat = round(mean spacing, 2) (:365) and the row weights
are float64 expressions cast to float32 (:703-704).  More than 65 536 timesteps: until round 5 the one-call pass left its fused form
there (per-pass filter launches, no k_compact_init); now it is the same pass as at 2707 steps, which these tests pin.

  (a) embedding: background except four windows that hold CESM-grid cases the C oracle can run -- one in front of step 65 536, one
      across it, one past it, one past step 400 000 (the last member).  `flag` inside every window = the oracle's, ids included
      (offset by the labels of the windows in front: scipy numbers over the whole slab), zero elsewhere;
  (b) the device-generated configs[4] slab (every member its own seed): the pass stays on its fused form (no host hand-off);
      device-side properties (flag inside the mask of a float64 compare, ids in range, distinct ids = n_tracked, nothing below
      `persistence`); one call against 8 time shards of 54 750 steps (the layout configs[4] names): equal position-weighted
      checksums per shard and equal n_tracked.
Needs ~250 GB of the 288 GB of HBM."""
import ctypes as C
import threading

import numpy as np
import pytest

from contrack_amd import _native, synth
from contrack_amd.contrack import row_weights
from contrack_amd.dist import shard_bounds

pytestmark = pytest.mark.gpu

T, NY, NX = 438000, 192, 288
MEMBER = 10950
PLANE = NY * NX
PERSISTENCE = 5


def _off(p, nbytes):
    return C.c_void_p(p.value + int(nbytes))


def cesm_latitudes(ny=NY):
    """Gaussian-like float64 latitudes: irregular spacing (tests/golden/make_golden.py::cesm_grid, the `cesm_like` golden)"""
    k = np.arange(ny)
    return (90.0 - 180.0 * (k + 0.5) / ny + 0.3 * np.sin(np.pi * k / (ny - 1))).astype(np.float64)


def cesm_weights():
    lat = cesm_latitudes()
    dlat = np.float64(round(float(np.abs(np.diff(lat)).mean()), 2))          # contrack.py:365 (force=True)
    return row_weights(lat, dlat, np.float64(360.0 / NX))


@pytest.fixture(scope="module")
def big():
    """the two 96.9 GB device slabs (owned by a handle that never tracks: its workspace stays empty)"""
    if _native.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the GPU box")
    mem = _native.Tracker(0)
    d_in = d_out = None
    try:
        d_in = mem.malloc(T * PLANE * 4)
        d_out = mem.malloc(T * PLANE * 4)
    except MemoryError as e:
        if d_in is not None:
            mem.free(d_in)
        mem.close()
        pytest.fail("the configs[4] slab needs 2 x 96.9 GB of device memory: %s" % e)
    yield mem, d_in, d_out, cesm_weights()
    mem.free(d_in)
    mem.free(d_out)
    mem.close()


def test_weights_are_the_references_force_path():
    """float64 latitudes -> float64 expression -> float32 (contrack.py:703-704); the oracle's row_weights agrees"""
    import oracle
    lat = cesm_latitudes()
    assert np.unique(np.round(np.abs(np.diff(lat)), 6)).size > 1              # irregular: the reference needs force=True
    dlat = np.float64(round(float(np.abs(np.diff(lat)).mean()), 2))
    assert dlat == np.float64(0.94)
    w = cesm_weights()
    assert w.dtype == np.float32 and np.array_equal(w, oracle.row_weights(lat, dlat, np.float64(360.0 / NX)))


def test_embedded_windows_before_across_and_past_65536_steps_match_oracle(big, oracle_lib):
    mem, d_in, d_out, w = big
    thr_val = 160.0
    cases = {"A": (150, 7), "B": (96, 8)}
    want, nlab = {}, {}
    small = _native.Tracker(0)
    try:
        for k, (n, seed) in cases.items():
            a = np.full((n + 2, NY, NX), -1000.0, dtype=np.float32)
            a[1:-1] = synth.smooth_field(n, NY, NX, seed=seed)
            thr = oracle_lib.prepare_thresholds(thr_val, n + 2)
            f, nt = oracle_lib.run_contrack(a, thr, ">=", w, 0.5, PERSISTENCE, True)
            g, ng = small.track(a, thr, 0, w, 0.5, PERSISTENCE, True)
            assert np.array_equal(g, f) and ng == nt and nt > 0
            want[k] = (a, f, nt)
            nlab[k] = small.stats()["labels_3d"]
    finally:
        small.close()
    # (first step, case): in front of step 65 536, across it, past it, in the last member (past step 400 000), at the very end
    windows = [(30000, "A"), (65536 - 70, "A"), (70000, "B"), (400100, "A"), (T - 98, "B")]
    trk = _native.Tracker(0)
    try:
        mem.memset(d_in, 0, T * PLANE * 4)                                     # 0.0 < 160: background
        mem.memset(d_out, 0xff, T * PLANE * 4)                                  # (every pixel of the result must be WRITTEN)
        for t0, k in windows:
            mem.h2d(_off(d_in, t0 * PLANE * 4), want[k][0])
        thr = oracle_lib.prepare_thresholds(thr_val, T)
        n = trk.track_dev(d_in, T, NY, NX, thr, 0, w, 0.5, PERSISTENCE, True, d_out)
        st = trk.stats()
        assert st["fused_pass"] == 1 and st["off_fused_path_reason"] == 0, st
        assert n == sum(want[k][2] for _, k in windows)
        assert st["labels_3d"] == sum(nlab[k] for _, k in windows)
        base, nz_expected = 0, 0
        for t0, k in windows:
            a, f, nt = want[k]
            got = np.empty(f.shape, dtype=np.int32)
            mem.d2h(got, _off(d_out, t0 * PLANE * 4))
            expect = np.where(f > 0, f + np.int32(base), 0).astype(np.int32)      # scipy numbers over the whole slab
            assert np.array_equal(got, expect), ("window at step %d" % t0)
            base += nlab[k]
            nz_expected += int(np.count_nonzero(f))
        _, nz = mem.checksum_i32(d_out, T * PLANE, 0)
        assert nz == nz_expected                                               # zero everywhere else
        pr = trk.check_flag(d_in, d_out, T, NY, NX, thr, 0, PERSISTENCE, st["labels_3d"])
        assert pr["flag_outside_mask"] == 0 and pr["ids_out_of_range"] == 0 and pr["ids_below_persistence"] == 0
        assert pr["ids"] == n and pr["nonzero"] == nz_expected
    finally:
        trk.close()


def test_configs4_slab_fused_pass_properties_and_eight_shards(big):
    mem, d_in, d_out, w = big
    for m, tb in enumerate(range(0, T, MEMBER)):                                # 40 members, each its own field
        mem.synth_fill(_off(d_in, tb * PLANE * 4), min(MEMBER, T - tb), NY, NX, seed=100 + m)
    thr = np.full(T, np.float64(np.float32(160.0)))
    mem.memset(d_out, 0xff, T * PLANE * 4)
    trk = _native.Tracker(0)
    try:
        n_first = trk.track_dev(d_in, T, NY, NX, thr, 0, w, 0.5, PERSISTENCE, True, d_out)
        first = trk.stats()
        n_one = trk.track_dev(d_in, T, NY, NX, thr, 0, w, 0.5, PERSISTENCE, True, d_out)
        st = trk.stats()
        assert n_one == n_first and n_one > 100000, (n_first, n_one, st)
        # the same fused pass as at 2707 steps: one launch for all filter passes, device seam driver, no host hand-off
        assert st["fused_pass"] == 1 and st["off_fused_path_reason"] == 0 and st["host_path"] == 0, (first, st)
        assert first["fused_pass"] == 1, (first["off_fused_path_reason"], first["filter_passes"])                                   # ... on a fresh handle already (first-call sizing)
        pr = trk.check_flag(d_in, d_out, T, NY, NX, thr, 0, PERSISTENCE, st["labels_3d"])
        assert pr["flag_outside_mask"] == 0 and pr["ids_out_of_range"] == 0, pr
        assert pr["ids_below_persistence"] == 0 and pr["ids"] == n_one, (pr, n_one)
        assert 0.02 < pr["nonzero"] / float(T * PLANE) < 0.2, pr
    finally:
        trk.close()                                                              # (its tables: ~40 GB the eight shard handles need)
    world = 8
    bounds = shard_bounds(T, world)
    assert all(b - a == 54750 for a, b in bounds)
    ref = [mem.checksum_i32(_off(d_out, a * PLANE * 4), (b - a) * PLANE, a * PLANE) for a, b in bounds]
    mem.memset(d_out, 0xff, T * PLANE * 4)
    hs = [_native.Tracker(0) for _ in range(world)]
    group = _native.CommGroup(world)
    comms = [_native.Comm.local(hs[r], group, r) for r in range(world)]
    res, err = [None] * world, [None] * world

    def work(r):
        a, b = bounds[r]
        try:
            res[r] = hs[r].track_sharded_dev(comms[r], _off(d_in, a * PLANE * 4), b - a, a, T, NY, NX, thr[a:b].copy(), 0, w, 0.5, PERSISTENCE, True,
                                             _off(d_out, a * PLANE * 4))
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
    got = [mem.checksum_i32(_off(d_out, a * PLANE * 4), (b - a) * PLANE, a * PLANE) for a, b in bounds]
    assert got == ref


def test_small_plane_variant_and_its_remainder_833_to_1024_runs(big, oracle_lib):
    """Long shards of small planes (ny <= 256, <= 960 mask words, more than 65 536 steps) are labelled by the 20 KB variant of the 2-D
    labelling kernel, which carries 832 runs per plane; planes with 833 .. 1024 runs go to the 1024-run variant behind it (round 5).
    A 66 000-step slab (the head of the big buffers) that is background except a window of speckled planes with ~900 runs each."""
    mem, d_in, d_out, w = big
    T2, n, t0 = 66000, 6, 41000
    rng = np.random.default_rng(5)
    speck = rng.random((NY, NX)) < 900.0 / PLANE
    a = np.full((n + 2, NY, NX), -1000.0, dtype=np.float32)
    for k in range(1, n + 1):
        m = speck ^ (rng.random((NY, NX)) < 20.0 / PLANE)                      # a few pixels come and go
        a[k][m] = 200.0
    fg = a[1:-1] >= 160.0
    runs = (fg & ~np.concatenate([np.zeros((n, NY, 1), bool), fg[:, :, :-1]], axis=2)).sum(axis=(1, 2))
    assert runs.max() <= 1024 and runs.min() > 832, runs
    thr_small = oracle_lib.prepare_thresholds(160.0, n + 2)
    want, nw = oracle_lib.run_contrack(a, thr_small, ">=", w, 0.5, 2, True)
    assert nw > 100
    trk = _native.Tracker(0)
    try:
        mem.memset(d_in, 0, T2 * PLANE * 4)
        mem.memset(d_out, 0xff, T2 * PLANE * 4)
        mem.h2d(_off(d_in, t0 * PLANE * 4), a)
        thr = oracle_lib.prepare_thresholds(160.0, T2)
        for _ in range(2):                                                      # (the second call launches the variants speculatively)
            ng = trk.track_dev(d_in, T2, NY, NX, thr, 0, w, 0.5, 2, True, d_out)
            st = trk.stats()
            assert 832 < st["max_runs_per_step"] <= 1024, st
            assert ng == nw
            got = np.empty(want.shape, dtype=np.int32)
            mem.d2h(got, _off(d_out, t0 * PLANE * 4))
            assert np.array_equal(got, want)
            assert mem.checksum_i32(d_out, T2 * PLANE, 0)[1] == int(np.count_nonzero(want))
            mem.memset(d_out, 0xff, T2 * PLANE * 4)
    finally:
        trk.close()
