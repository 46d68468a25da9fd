"""The product's time-sharded path -- ctk_track_sharded_* (contrack_amd/csrc/ctk_sharded.hip) -- on ONE GPU: N handles play N
ranks, one host thread each, joined by an in-process communicator group (ctk_comm_init_local).  Everything but the transport
is what runs with RCCL on N GPUs: halo exchange, boundary-coupled overlap filter, scipy ids across shard boundaries, shared /
local seam merges, extent exchange, exact ties on rounded area sums.  Bit-exact against the reference goldens, the oracle and
the one-call result -- no case is skipped."""
import numpy as np
import pytest

import golden_util
from contrack_amd import _native, synth
from shard_inproc import sharded_threads
from test_gpu_parity import _edge_case, _random_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handles():
    hs = [_native.Tracker(0) for _ in range(9)]
    yield hs
    for h in hs:
        h.close()


def _cuts(T, n):
    c = sorted(set(int(round(T * k / n)) for k in range(n + 1)))
    return c if len(c) == n + 1 else None


@pytest.mark.parametrize("name", golden_util.case_names())
def test_sharded_matches_reference_golden(handles, name):
    """every golden (the reference's own outputs), split into 2, 3 and 5 time shards -- incl. f64pole_blocky*: exact ties on
    components with pole-row pixels of a float64-latitude grid, resolved with numpy-order sums inside the sharded path"""
    g = golden_util.load(name)
    T = g["anom"].shape[0]
    op = _native.CMP_OPS[g["gorl"]]
    want_n = len(np.unique(g["flag"])) - 1
    for n in (2, 3, 5):
        cuts = _cuts(T, n)
        if cuts is None:
            continue
        f, nt, st = sharded_threads(handles[:n], g["anom"], g["thr"], op, g["wrow"], g["overlap"], g["persistence"], g["twosided"], cuts)
        assert np.array_equal(f, g["flag"]) and nt == want_n, (name, cuts)
        if name == "f64pole_blocky":
            assert sum(s["exact_fixups"] for s in st) > 0


@pytest.mark.parametrize("i", range(48))
def test_random_shard_boundaries(handles, oracle_lib, i):
    """random fields / parameters, random cuts (one-step shards included) against the one-call result and the oracle"""
    a, thr, gorl, ov, pers, two = _random_case(i) if i % 3 else _edge_case(i)
    T, ny, nx = a.shape
    if T < 2:
        pytest.skip("single step: nothing to shard")
    rng = np.random.default_rng(7000 + i)
    n = int(rng.integers(2, min(6, T) + 1))
    cuts = [0] + [int(v) for v in np.sort(rng.choice(np.arange(1, T), size=n - 1, replace=False))] + [T]
    lat = np.linspace(90, -90, ny).astype(np.float32)
    w = oracle_lib.row_weights(lat, np.float32(180.0 / max(ny - 1, 1)), np.float32(360.0 / nx))
    thrv = oracle_lib.prepare_thresholds(thr, T)
    op = _native.CMP_OPS[gorl]
    want, nw = handles[6].track(a, thrv, op, w, ov, pers, two)
    got, ng, _ = sharded_threads(handles[:n], a, thrv, op, w, ov, pers, two, cuts)
    assert np.array_equal(got, want) and ng == nw, cuts
    if i % 4 == 0:
        ow, on = oracle_lib.run_contrack(a, thrv, gorl, w, ov, pers, two)
        assert np.array_equal(got, ow) and ng == on


def test_sharded_025deg_grid(handles, oracle_lib):
    """BASELINE configs[3] grid (721 x 1440), three shards, one of a single step, persistence 3 -- against the oracle"""
    T, ny, nx = 14, 721, 1440
    a = synth.smooth_field(T, ny, nx, seed=31)
    lat = np.linspace(90, -90, ny).astype(np.float32)
    w = oracle_lib.row_weights(lat, np.float32(0.25), np.float32(0.25))
    thr = oracle_lib.prepare_thresholds(160.0, T)
    want, nw = oracle_lib.run_contrack(a, thr, ">=", w, 0.5, 3, True)
    got, ng, _ = sharded_threads(handles[:3], a, thr, 0, w, 0.5, 3, True, [0, 6, 7, T])
    assert np.array_equal(got, want) and ng == nw and nw > 3


def test_sharded_cesm_grid_irregular_f64_latitudes(handles, oracle_lib):
    """BASELINE configs[4] grid (192 x 288): float64 Gaussian-like latitudes (the reference needs set_up(force=True); dlat =
    round(mean spacing, 2), contrack.py:357-370), four shards incl. a one-step shard, concatenated 'members'"""
    T, ny, nx = 36, 192, 288
    a = np.concatenate([synth.smooth_field(T // 2, ny, nx, seed=s) for s in (41, 42)], axis=0)      # two members on the time axis
    k = np.arange(ny)
    lat = (90.0 - 180.0 * (k + 0.5) / ny + 0.3 * np.sin(np.pi * k / (ny - 1))).astype(np.float64)
    dlat = np.float64(round(float(np.abs(np.diff(lat)).mean()), 2))
    w = oracle_lib.row_weights(lat, dlat, np.float64(360.0 / nx))
    thr = oracle_lib.prepare_thresholds(150.0, T)
    want, nw = oracle_lib.run_contrack(a, thr, ">=", w, 0.5, 3, True)
    got, ng, _ = sharded_threads(handles[:4], a, thr, 0, w, 0.5, 3, True, [0, 11, 18, 19, T])
    assert np.array_equal(got, want) and ng == nw and nw > 3


def test_sharded_float64_slab(handles):
    g = golden_util.load("thr_vector")
    a64 = g["anom"].astype(np.float64)
    op = _native.CMP_OPS[g["gorl"]]
    want, nw = handles[6].track(a64, g["thr"], op, g["wrow"], g["overlap"], g["persistence"], g["twosided"], f64=True)
    got, ng, _ = sharded_threads(handles[:3], a64, g["thr"], op, g["wrow"], g["overlap"], g["persistence"], g["twosided"], [0, 13, 27, 40], f64=True)
    assert np.array_equal(got, want) and ng == nw


def test_long_slab_many_seam_operations(handles):
    """thousands of seam operations, op chains across shard boundaries: eight shards of a 2707-step slab (2 deg grid)"""
    from contrack_amd.contrack import row_weights
    T, ny, nx = 2707, 91, 180
    anom = synth.smooth_field(T, ny, nx, seed=11)
    lat, _ = synth.grid(ny, nx)
    wrow = row_weights(lat, np.float32(2.0), np.float32(2.0))
    thr = np.full(T, 120.0)
    want, nw = handles[6].track(anom, thr, 0, wrow, 0.5, 3, True)
    assert handles[6].stats()["seam_ops"] > 300
    cuts = [0, 300, 301, 900, 1500, 2000, 2706, T]
    got, ng, st = sharded_threads(handles[:7], anom, thr, 0, wrow, 0.5, 3, True, cuts)
    assert np.array_equal(got, want) and ng == nw
    assert st[0]["shared_seam_rows"] < st[0]["seam_rows_to_driver"] + sum(s["seam_rows_to_driver"] for s in st)      # most groups stay local


def test_extent_exchange_forms_and_x4_speculation(handles):
    """the fused extent + count exchange in its one-workgroup form (list of shared ids longer than its LDS copy: forced by the
    mailbox hook) gives the same result as the 32-workgroup form; the stats say whether the boundary records of the 3-D
    labelling travelled with the filter's last exchange (several shards: from the second round on)"""
    g = golden_util.load("busy_s1")
    T = g["anom"].shape[0]
    op = _native.CMP_OPS[g["gorl"]]
    cuts = [0, T // 3, 2 * T // 3, T]
    args = (g["anom"], g["thr"], op, g["wrow"], g["overlap"], g["persistence"], g["twosided"], cuts)
    f0, n0, st0 = sharded_threads(handles[:3], *args)
    try:
        for h in handles[:3]:
            h.debug_set_mailbox(0, 1)
        f1, n1, st1 = sharded_threads(handles[:3], *args)
    finally:
        for h in handles[:3]:
            h.debug_set_mailbox(0, 0)
    assert np.array_equal(f0, g["flag"]) and np.array_equal(f1, g["flag"]) and n0 == n1 == len(np.unique(g["flag"])) - 1
    for s in st0:
        assert s["x4_speculated"] == (1 if s["filter_rounds"] > 1 else 0), s


def test_bench_slab_in_eight_shards(handles):
    """BASELINE configs[1] complete (2707 x 181 x 360, the bench slab, compared with the oracle in test_gpu_parity.py) cut into the
    eight time shards an 8-GPU node would hold: the strong-scaling layout of `bench.py --gpus 8 --scaling strong`"""
    from contrack_amd.contrack import row_weights
    from contrack_amd.dist import shard_bounds
    T, ny, nx = 2707, 181, 360
    anom = synth.smooth_field(T, ny, nx, seed=0)
    lat, _ = synth.grid(ny, nx)
    wrow = row_weights(lat, np.float32(1.0), np.float32(1.0))
    thr = np.full(T, 160.0)
    want, nw = handles[8].track(anom, thr, 0, wrow, 0.5, 5, True)
    assert nw == 3305
    b = shard_bounds(T, 8)
    got, ng, _ = sharded_threads(handles[:8], anom, thr, 0, wrow, 0.5, 5, True, [x[0] for x in b] + [T])
    assert np.array_equal(got, want) and ng == nw


def test_025deg_persistence_20_in_four_shards(handles):
    """BASELINE configs[2] / [3] parameters (721 x 1440, persistence 20): tracks that live across three shard boundaries"""
    from contrack_amd.contrack import row_weights
    T, ny, nx = 64, 721, 1440
    anom = synth.smooth_field(T, ny, nx, seed=5)
    lat, _ = synth.grid(ny, nx)
    wrow = row_weights(lat, np.float32(0.25), np.float32(0.25))
    thr = np.full(T, 160.0)
    want, nw = handles[8].track(anom, thr, 0, wrow, 0.5, 20, True)
    got, ng, _ = sharded_threads(handles[:4], anom, thr, 0, wrow, 0.5, 20, True, [0, 16, 32, 48, T])
    assert np.array_equal(got, want) and ng == nw and nw > 0


def test_staged_import_between_two_first_shard_passes_on_one_handle(handles):
    """one handle plays: the only / first time shard (its halo header is zeroed once and remembered), then rank 1 of the STAGED
    protocol (ctk_shard_halo_import writes a v1 blob over that header), then a first shard again -- the remembered zero header
    must not survive the import (advisor finding, round 3)"""
    from shard_inproc import sharded
    g = golden_util.load("busy_s1")
    T = g["anom"].shape[0]
    op = _native.CMP_OPS[g["gorl"]]
    want_n = len(np.unique(g["flag"])) - 1
    args = (g["anom"], g["thr"], op, g["wrow"], g["overlap"], g["persistence"], g["twosided"])
    h = handles[0]
    for rep in range(2):
        f, n, _ = sharded_threads([h], *args, [0, T])
        assert np.array_equal(f, g["flag"]) and n == want_n, ("first shard alone", rep)
        f, n, _ = sharded([handles[1], h], *args, [0, T // 2, T])                  # h imports the halo of handles[1]
        assert np.array_equal(f, g["flag"]) and n == want_n, ("staged", rep)
        f, n, _ = sharded_threads([h, handles[2]], *args, [0, T // 3, T])            # h is rank 0 of two
        assert np.array_equal(f, g["flag"]) and n == want_n, ("rank 0 of two", rep)


def test_device_seam_driver_on_shards_and_its_collective_fallback(handles):
    """the shard path drives the seam merges of its own clusters on the device (no host hand-off behind the boundary resolution) and
    only the clusters shared between shards on the host; a rank whose device driver cannot hold a cluster (tables cut down by the
    test hook) says so through the extent exchange and EVERY rank repeats X5 .. X7 host-driven -- same result"""
    from contrack_amd.contrack import row_weights
    T, ny, nx = 400, 91, 180
    anom = synth.smooth_field(T, ny, nx, seed=12)
    lat, _ = synth.grid(ny, nx)
    wrow = row_weights(lat, np.float32(2.0), np.float32(2.0))
    thr = np.full(T, 120.0)
    want, nw = handles[6].track(anom, thr, 0, wrow, 0.5, 3, True)
    assert handles[6].stats()["seam_ops"] > 20
    cuts = [0, 90, 91, 250, T]
    got, ng, st = sharded_threads(handles[:4], anom, thr, 0, wrow, 0.5, 3, True, cuts)
    assert np.array_equal(got, want) and ng == nw
    assert all(s["fused_pass"] == 1 and not (s["off_fused_path_reason"] & 16) for s in st), st
    try:
        handles[2].debug_set_seam_caps(1, 1)                   # rank 2's device driver holds one label per cluster: every real cluster poisons it
        got, ng, st = sharded_threads(handles[:4], anom, thr, 0, wrow, 0.5, 3, True, cuts)
        assert np.array_equal(got, want) and ng == nw
        assert all(s["fused_pass"] == 0 and (s["off_fused_path_reason"] & 16) for s in st), st
        got, ng, st = sharded_threads(handles[:4], anom, thr, 0, wrow, 0.5, 3, True, cuts)       # rank 2 stays host-driven on this grid, the others try again
        assert np.array_equal(got, want) and ng == nw
    finally:
        handles[2].debug_set_seam_caps(0, 0)


def test_rccl_world_of_one(oracle_lib):
    """the RCCL transport itself (librccl.so dlopen'ed, ncclCommInitRank, ncclAllGather) with one rank: what a one-GPU box can run"""
    g = golden_util.load("busy_s0")
    T, ny, nx = g["anom"].shape
    with _native.Tracker(0) as t:
        comm = _native.Comm.rccl(t, _native.comm_unique_id(), 0, 1)
        try:
            d_in, d_out = t.malloc(g["anom"].nbytes), t.malloc(g["anom"].nbytes)
            t.h2d(d_in, g["anom"])
            n = t.track_sharded_dev(comm, d_in, T, 0, T, ny, nx, g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"], d_out)
            f = np.empty((T, ny, nx), np.int32)
            t.d2h(f, d_out)
            comm.barrier()
            assert comm.allgather(np.array([3.5])).tolist() == [[3.5]]
            assert np.array_equal(f, g["flag"]) and n == len(np.unique(g["flag"])) - 1
            t.free(d_in); t.free(d_out)
        finally:
            comm.close()


def test_bad_shard_arguments(handles):
    g = golden_util.load("T3")
    group = _native.CommGroup(1)
    comm = _native.Comm.local(handles[0], group, 0)
    d = handles[0].malloc(g["anom"].nbytes)
    try:
        with pytest.raises(ValueError):          # the only rank must own the whole time axis
            handles[0].track_sharded_dev(comm, d, 2, 0, 3, 31, 60, g["thr"][:2], 0, g["wrow"], 0.5, 2, True, d)
        with pytest.raises(ValueError):          # a rank without a timestep
            handles[0].track_sharded_dev(comm, d, 0, 0, 0, 31, 60, g["thr"][:0], 0, g["wrow"], 0.5, 2, True, d)
    finally:
        handles[0].free(d)
        comm.close()
        group.close()
