"""HIP path (through the C ABI, contrack_amd/_native.py) against the reference goldens and the CPU
oracle.  Bit-exact: ids are the reference's ids (identity permutation)."""
import numpy as np
import pytest

import cpu_tables
import golden_util
from contrack_amd import _native, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def trk():
    if _native.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the GPU box")
    t = _native.Tracker(0)
    yield t
    t.close()


@pytest.mark.parametrize("name", golden_util.case_names())
def test_hip_matches_reference_golden(trk, name):
    g = golden_util.load(name)
    flag, n = trk.track(g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
    assert flag.dtype == np.int32
    assert np.array_equal(flag, g["flag"])
    assert n == len(np.unique(g["flag"])) - 1


@pytest.mark.parametrize("name", golden_util.case_names())
def test_extent_kernel_with_sixteen_timesteps_per_workgroup(name):
    """k_extent_blk (round 6: the ids' time extents reduced in LDS per sixteen timesteps before they touch memory) serves shards of more
    than 2048 timesteps by default; forced here (ctk_debug_set_small_threads extent = 1024) on every golden -- chain_* hold the complex
    components whose pixels are folded one by one -- through the dense device path and through the host entry."""
    g = golden_util.load(name)
    op = _native.CMP_OPS[g["gorl"]]
    t = _native.Tracker(0)
    try:
        _native.check(_native.lib().ctk_debug_set_small_threads(t.handle, 1024, 0, 0))
        flag, n = t.track(g["anom"], g["thr"], op, g["wrow"], g["overlap"], g["persistence"], g["twosided"])
        assert np.array_equal(flag, g["flag"]) and n == len(np.unique(g["flag"])) - 1
        T, ny, nx = g["anom"].shape
        d_in, d_out = t.malloc(g["anom"].nbytes), t.malloc(T * ny * nx * 4)
        try:
            t.h2d(d_in, np.ascontiguousarray(g["anom"]))
            for _ in range(2):                               # (the second call runs speculatively on the first one's capacities)
                n2 = t.track_dev(d_in, T, ny, nx, g["thr"], op, g["wrow"], g["overlap"], g["persistence"], g["twosided"], d_out)
                out = np.empty((T, ny, nx), dtype=np.int32)
                t.d2h(out, d_out)
                assert np.array_equal(out, g["flag"]) and n2 == n
        finally:
            t.free(d_in)
            t.free(d_out)
    finally:
        t.close()


def _staged(trk, anom, thr, op, wrow):
    T, ny, nx = anom.shape
    d = trk.malloc(anom.nbytes)
    try:
        trk.h2d(d, anom)
        trk.shard_label2d(d, T, ny, nx, thr, op, wrow, False)
        mask = trk.debug_mask(T, ny, nx)
        lab_nw = trk.debug_label2d(T, ny, nx, True)
        lab_m = trk.debug_label2d(T, ny, nx, False)
        trk.shard_overlap()
        blob = trk.shard_tables()
    finally:
        trk.free(d)
    return mask, lab_nw, lab_m, blob


@pytest.mark.parametrize("name", ["refslab_two", "syn2deg_s0", "busy_s1", "noise", "odd_65x130", "odd_9x65", "nan_speckle",
                                  "thr_vector", "all_fg", "all_bg", "chain_a", "cesm_like"])
def test_staged_outputs_match_oracle(trk, oracle_lib, name):
    """threshold mask, scipy-numbered 2-D labels before/after the seam merge (contrack.py:684-698), and the
    component / pair / seam tables."""
    g = golden_util.load(name)
    op = _native.CMP_OPS[g["gorl"]]
    mask, lab_nw, lab_m, blob = _staged(trk, g["anom"], g["thr"], op, g["wrow"])
    omask = oracle_lib.threshold_mask(g["anom"], g["thr"], g["gorl"])
    assert np.array_equal(mask, omask)
    olab, _ = oracle_lib.label(omask, 0)
    assert np.array_equal(lab_nw, olab)
    _, _, stage = oracle_lib.run_contrack(g["anom"], g["thr"], g["gorl"], g["wrow"], g["overlap"], g["persistence"], g["twosided"],
                                          return_stage=True)
    assert np.array_equal(lab_m, stage)
    wlo, whi, wshift, lb = _native.weights_to_limbs(g["wrow"], npix=omask.shape[1] * omask.shape[2], with_bits=True)
    ref = cpu_tables.parse_blob(cpu_tables.pack_blob(cpu_tables.build_tables(omask.astype(bool), wlo, whi), wshift, False, limb_bits=lb))
    got = cpu_tables.parse_blob(blob)
    assert got["T"] == ref["T"] and got["wshift"] == ref["wshift"]
    assert np.array_equal(got["ncomp"], ref["ncomp"])
    assert np.array_equal(got["mrep"], ref["mrep"])
    assert np.array_equal(got["box"], ref["box"])
    assert np.array_equal(got["area"], ref["area"])
    assert got["pairs"] == ref["pairs"]
    assert got["seams"] == ref["seams"]


CASES_VS_ORACLE = [
    # (T, ny, nx, seed, kind, threshold, gorl, overlap, persistence, twosided)
    (96, 181, 360, 3, "smooth", 160.0, ">=", 0.5, 5, True),          # BASELINE config-1 parameters
    (64, 181, 360, 4, "smooth", 150.0, ">", 0.5, 5, False),
    (6, 721, 1440, 5, "smooth", 160.0, ">=", 0.5, 2, True),          # 0.25 deg grid: 23 words per row
    (8, 181, 360, 6, "noise", 0.8, ">=", 0.5, 2, True),              # ~10^4 runs per step: global-memory labelling variant
    (6, 181, 360, 12, "noise", 2.2, ">=", 0.5, 2, True),             # ~900 runs per step: beyond the 768 of the 20 KB labelling variant for 181 x 360 planes
    (5, 192, 288, 7, "smooth", 160.0, ">=", 0.5, 2, True),           # CESM grid
    (1, 91, 180, 8, "smooth", 150.0, ">=", 0.5, 1, True),            # T = 1 (the reference cannot even set up)
    (40, 91, 4200, 9, "smooth", 150.0, ">=", 0.5, 3, True),          # more than 64 words per row
    (12, 2100, 64, 10, "smooth", 150.0, ">=", 0.5, 2, True),         # more rows than the LDS row table
    # the BASELINE.json configurations at their own sizes / parameters (the C oracle needs a few seconds for each):
    (2707, 181, 360, 0, "smooth", 160.0, ">=", 0.5, 5, True),        # configs[1] complete: the bench slab itself, 3305 tracks
    (64, 721, 1440, 5, "smooth", 160.0, ">=", 0.5, 20, True),        # configs[2]'s grid with its persistence of 20 steps
    (240, 192, 288, 7, "smooth", 160.0, ">=", 0.5, 5, True),         # configs[4]'s grid (CESM 0.9 x 1.25 deg), eight months daily
    # more than 65 536 timesteps: the fused pass without k_compact_init (`!fz_init`), per-pass filter launches beyond 60 000 steps,
    # the grid-stride form of the device seam driver
    (70000, 8, 16, 11, "smooth", 120.0, ">=", 0.5, 3, True),
]


@pytest.mark.parametrize("case", CASES_VS_ORACLE, ids=lambda c: "%dx%dx%d_%s" % (c[0], c[1], c[2], c[4]))
def test_hip_matches_oracle(trk, oracle_lib, case):
    T, ny, nx, seed, kind, thr, gorl, ov, pers, two = case
    if kind == "noise":
        a = np.random.default_rng(seed).standard_normal((T, ny, nx)).astype(np.float32)
    else:
        a = synth.smooth_field(T, ny, nx, seed=seed)
    lat = np.linspace(90, -90, ny).astype(np.float32)
    w = oracle_lib.row_weights(lat, np.float32(180.0 / (ny - 1)), np.float32(360.0 / nx))
    thrv = oracle_lib.prepare_thresholds(thr, T)
    want, nw = oracle_lib.run_contrack(a, thrv, gorl, w, ov, pers, two)
    got, ng = trk.track(a, thrv, _native.CMP_OPS[gorl], w, ov, pers, two)
    assert np.array_equal(got, want)
    assert ng == nw


def _random_case(i):
    """seeded random shape / field / parameters: (anom, threshold, gorl, overlap, persistence, twosided)"""
    rng = np.random.default_rng(1000 + i)
    T = int(rng.integers(2, 48))
    ny = int(rng.choice([19, 37, 46, 61, 91, 121]))
    nx = int(rng.choice([24, 48, 70, 72, 96, 130, 144, 180, 256, 300]))
    kind = rng.choice(["smooth", "smooth", "smooth", "blocky", "noise"])
    if kind == "noise":
        a = rng.standard_normal((T, ny, nx)).astype(np.float32)
        thr = float(rng.choice([0.6, 1.0, 1.4]))
    elif kind == "blocky":
        # coarse random field repeated over 3x3 pixels and two steps: plateaus, exact overlap ties, seam-hugging blobs
        c = rng.standard_normal(((T + 1) // 2, (ny + 2) // 3, (nx + 2) // 3)).astype(np.float32)
        a = np.repeat(np.repeat(np.repeat(c, 2, axis=0), 3, axis=1), 3, axis=2)[:T, :ny, :nx].copy()
        thr = float(rng.choice([0.3, 0.8]))
    else:
        a = synth.smooth_field(T, ny, nx, seed=int(rng.integers(1 << 30)), sigma_deg=float(rng.choice([6.0, 12.0, 20.0])))
        thr = float(rng.choice([100.0, 130.0, 160.0]))
    gorl = str(rng.choice([">=", ">", "<=", "<"]))
    if gorl in ("<=", "<"):
        a = (-a + np.float32(2 * 35.0 if kind == "smooth" else 0.0)).astype(np.float32)      # same coverage from below
        thr = -thr + (70.0 if kind == "smooth" else 0.0)
    overlap = float(rng.choice([0.0, 0.25, 0.5, 0.5, 0.85, 1.0]))
    persistence = int(rng.integers(1, 6))
    twosided = bool(rng.integers(0, 2))
    return a, thr, gorl, overlap, persistence, twosided


def _edge_case(i):
    """tiny grids, T = 1..4, NaNs, thresholds that equal data values, zonal bands, per-step thresholds"""
    rng = np.random.default_rng(50000 + i)
    T = int(rng.integers(1, 9)); ny = int(rng.integers(2, 24)); nx = int(rng.choice([4, 5, 8, 12, 63, 64, 65, 100, 128, 129]))
    levels = rng.integers(-3, 4, size=(T, ny, nx)).astype(np.float32)            # few distinct values: ties with the threshold
    if rng.random() < 0.5:
        levels = np.repeat(np.repeat(levels[:, ::2, ::2], 2, axis=1), 2, axis=2)[:, :ny, :nx].copy()
    if rng.random() < 0.3:
        levels[:, int(rng.integers(0, ny)), :] = 3                                   # a zonal band: wraps around the seam
    if rng.random() < 0.3:
        levels[rng.random(levels.shape) < 0.05] = np.nan
    thr = rng.integers(-1, 3, size=T).astype(np.float64) if rng.random() < 0.5 else float(rng.integers(-1, 3))
    return levels, thr, str(rng.choice([">=", ">", "<=", "<"])), float(rng.choice([0.0, 0.5, 1.0, 0.3])), int(rng.integers(1, 4)), bool(rng.integers(0, 2))


@pytest.mark.parametrize("i", range(40))
def test_randomized_edge_cases_against_oracle(trk, oracle_lib, i):
    a, thr, gorl, ov, pers, two = _edge_case(i)
    T, ny, nx = a.shape
    lat = np.linspace(90, -90, ny).astype(np.float32)
    w = oracle_lib.row_weights(lat, np.float32(180.0 / max(ny - 1, 1)), np.float32(360.0 / nx))
    thrv = oracle_lib.prepare_thresholds(thr, T)
    want, nw = oracle_lib.run_contrack(a, thrv, gorl, w, ov, pers, two)
    got, ng = trk.track(a, thrv, _native.CMP_OPS[gorl], w, ov, pers, two)
    assert np.array_equal(got, want) and ng == nw and trk.stats()["ambiguous_decisions"] == 0


@pytest.mark.parametrize("i", range(36))
def test_randomized_against_oracle(trk, oracle_lib, i):
    """random grids (incl. nx not a multiple of 4 / 64), fields, comparators, overlaps, persistences: bit-exact ids"""
    a, thr, gorl, ov, pers, two = _random_case(i)
    T, ny, nx = a.shape
    lat = np.linspace(90, -90, ny).astype(np.float32)
    w = oracle_lib.row_weights(lat, np.float32(180.0 / (ny - 1)), np.float32(360.0 / nx))
    thrv = oracle_lib.prepare_thresholds(thr, T)
    want, nw = oracle_lib.run_contrack(a, thrv, gorl, w, ov, pers, two)
    got, ng = trk.track(a, thrv, _native.CMP_OPS[gorl], w, ov, pers, two)
    st = trk.stats()
    # Case 6 (3x3-blocky field, overlap 1.0) has exact ties on components that touch a pole row, whose area sums do not
    # fit float64: the device flags them and re-evaluates exactly those decisions with numpy-order sums (per-row pixel counts
    # from the masks and run tables -> the raster-order weight sequence np.sum sees), staying on the device.
    assert st["ambiguous_decisions"] == 0
    assert i != 6 or (st["exact_fixups"] > 0 and st["host_path"] == 0 and st["off_fused_path_reason"] == 4)
    assert np.array_equal(got, want) and ng == nw           # (30 of the 36 cases track something; 6 filter everything out)


@pytest.mark.parametrize("name", ["syn2deg_s0", "busy_s0", "chain_a", "chain_b", "chain_c", "noise", "syn2deg_fwd", "all_fg", "T2"])
def test_host_and_device_resolver_agree(trk, name):
    """ctk_track_* with the device resolver (default) and with the GPU-free host resolver give the same flag"""
    g = golden_util.load(name)
    args = (g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
    try:
        trk.set_device_resolve(False)
        f_host, n_host = trk.track(*args)
    finally:
        trk.set_device_resolve(True)
    f_dev, n_dev = trk.track(*args)
    assert np.array_equal(f_host, g["flag"]) and np.array_equal(f_dev, g["flag"]) and n_host == n_dev


@pytest.mark.parametrize("seed,T", [(3, 1500), (11, 2707)])
def test_host_and_device_resolver_agree_on_long_slabs(seed, T):
    """thousands of seam operations, labels that are `hi` of several of them (op chains), candidate tables beyond
    the first mailbox size: device resolver (fresh handle and warmed-up handle) against the host resolver"""
    from contrack_amd import synth
    from contrack_amd.contrack import row_weights
    ny, nx = 91, 180
    anom = synth.smooth_field(T, ny, nx, seed=seed)
    lat, _ = synth.grid(ny, nx)
    wrow = row_weights(lat, np.float32(2.0), np.float32(2.0))
    args = (anom, np.full(T, 120.0), 0, wrow, 0.5, 3, True)
    with _native.Tracker(0) as t:
        t.set_device_resolve(False)
        f_host, n_host = t.track(*args)
    with _native.Tracker(0) as t:
        f_dev, n_dev = t.track(*args)
        st = t.stats()
        f_dev2, n_dev2 = t.track(*args)
    assert st["host_path"] == 0 and st["seam_ops"] > 300
    assert n_host == n_dev == n_dev2
    assert np.array_equal(f_host, f_dev) and np.array_equal(f_host, f_dev2)


@pytest.mark.parametrize("caps", [(2, 100000), (100000, 3), (1, 1)])
def test_resolver_mailbox_too_small(caps):
    """candidate records / label tables that do not fit the device-written mailbox travel by explicit copies"""
    with _native.Tracker(0) as t:
        t.debug_set_mailbox(*caps)
        for name in ("chain_a", "chain_b", "syn2deg_s0", "busy_s0"):
            g = golden_util.load(name)
            f, n = t.track(g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
            assert t.stats()["host_path"] == 0 and t.stats()["seam_rows_to_driver"] > 3          # more than the caps hold
            assert np.array_equal(f, g["flag"]), name


def test_rare_paths_are_exercised(oracle_lib):
    """white noise at 181x360 drives every capacity path at once: the global-memory labelling variant (> 4096 runs
    per step), co-occurrence records that bypass the LDS hash table, and a pair table that has to be regrown (host
    resolver path) -- and the result still equals the oracle.  Then a long removal cascade: more filter passes than
    one round carries."""
    t = _native.Tracker(0)
    try:
        T, ny, nx = 6, 181, 360
        a = np.random.default_rng(99).standard_normal((T, ny, nx)).astype(np.float32)
        lat = np.linspace(90, -90, ny).astype(np.float32)
        w = oracle_lib.row_weights(lat, np.float32(1.0), np.float32(1.0))
        thr = oracle_lib.prepare_thresholds(0.8, T)
        want, nw = oracle_lib.run_contrack(a, thr, ">=", w, 0.5, 2, True)
        t.debug_set_pair_capacity(1000)                               # far too small: must be regrown
        got, ng = t.track(a, thr, 0, w, 0.5, 2, True)
        st = t.stats()
        assert np.array_equal(got, want) and ng == nw
        assert st["max_runs_per_step"] > 4096 and st["pair_table_regrows"] >= 1 and st["host_path"] == 1
        got2, _ = t.track(a, thr, 0, w, 0.5, 2, True)                 # table is large enough now: device resolver
        st2 = t.stats()
        assert np.array_equal(got2, want) and st2["host_path"] == 0 and st2["ungrouped_pairs"] > 0
        # several rounds of filter passes: convergence is only checked every `filter_round` passes.  The fused pass launches that
        # many, finds at its end that the filter had not converged, and the synchronous path repeats the resolution ...
        g = golden_util.load("syn2deg_s0")
        args = (g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
        t.set_filter_round(2)
        f, n = t.track(*args)
        st = t.stats()
        assert np.array_equal(f, g["flag"]) and st["host_path"] == 0 and st["fused_pass"] == 0
        assert st["filter_passes"] > 2 and st["filter_rounds"] == (st["filter_passes"] + 1) // 2
        # ... and launches more the next time, until the fused pass carries the whole cascade
        for _ in range(4):
            f, n = t.track(*args)
            assert np.array_equal(f, g["flag"])
            if t.stats()["fused_pass"]:
                break
        assert t.stats()["fused_pass"] == 1 and t.stats()["filter_passes"] > 2
        # the synchronous path alone, same bookkeeping
        t.set_fused(False)
        t.set_filter_round(2)
        f, n = t.track(*args)
        st = t.stats()
        assert np.array_equal(f, g["flag"]) and st["fused_pass"] == 0 and st["filter_rounds"] == (st["filter_passes"] + 1) // 2
    finally:
        t.close()


@pytest.mark.parametrize("name", ["chain_a", "chain_b", "chain_c", "busy_s1", "refslab_fwd", "noise"])
def test_fused_pass_and_its_fallbacks(name):
    """the one-call pass without a host hand-off (device seam driver: one wave per cluster of candidate labels, contrack.py:753-763)
    against the goldens; with the driver's per-cluster tables cut down to one label (an operation needs two) the clusters of these cases do not
    fit, the pass says so at its end and the synchronous path (host driver) repeats the resolution -- same result, and the grid
    stays on the host driver from then on"""
    g = golden_util.load(name)
    args = (g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
    want_n = len(np.unique(g["flag"])) - 1
    with _native.Tracker(0) as t:
        f, n = t.track(*args)
        st = t.stats()
        assert np.array_equal(f, g["flag"]) and n == want_n
        fused_first = st["fused_pass"]
        t.set_fused(False)
        f0, n0 = t.track(*args)
        assert np.array_equal(f0, g["flag"]) and n0 == want_n and t.stats()["fused_pass"] == 0
        if fused_first and st["seam_ops"] > 0:
            t.set_fused(True)
            t.debug_set_seam_caps(1, 1)
            f1, n1 = t.track(*args)
            assert np.array_equal(f1, g["flag"]) and n1 == want_n and t.stats()["fused_pass"] == 0
            f2, n2 = t.track(*args)                          # sticky: no second attempt on this grid
            assert np.array_equal(f2, g["flag"]) and t.stats()["fused_pass"] == 0


def test_bounded_inter_workgroup_waits_fall_back_instead_of_hanging():
    """the one-launch ("systolic") filter pass waits for the workgroup of the previous timesteps; HIP promises nothing about
    dispatch order, so every such wait is bounded: with the head of the chain made late (test hook) the pass still completes
    normally; with a head that NEVER publishes, the waits give up after the limit, the pass is marked invalid and the call
    repeats the resolution with one launch per filter pass -- an error-free, bit-exact result instead of a hung GPU -- and the
    handle stays on per-pass launches (contrack.py:706-742 is what both forms evaluate)"""
    import time
    g = golden_util.load("busy_s1")
    args = (g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
    want_n = len(np.unique(g["flag"])) - 1
    with _native.Tracker(0) as t:
        f, n = t.track(*args)
        assert np.array_equal(f, g["flag"]) and n == want_n and t.stats()["fused_pass"] == 1
        t.debug_set_spin(40.0, 1)                              # late by limit / 4: nobody gives up
        f, n = t.track(*args)
        st = t.stats()
        assert np.array_equal(f, g["flag"]) and n == want_n and st["fused_pass"] == 1 and not (st["off_fused_path_reason"] & 8), st
        t.debug_set_spin(20.0, 2)                              # never: every wave behind the head gives up after 20 ms
        t0 = time.time()
        f, n = t.track(*args)
        took = time.time() - t0
        st = t.stats()
        assert np.array_equal(f, g["flag"]) and n == want_n
        assert st["fused_pass"] == 0 and (st["off_fused_path_reason"] & 8), st
        assert took < 5.0, took
        f, n = t.track(*args)                                  # sticky: per-pass launches now, the stalled head no longer matters
        st = t.stats()
        assert np.array_equal(f, g["flag"]) and n == want_n and st["fused_pass"] == 1 and not (st["off_fused_path_reason"] & 8), st
        t.debug_set_spin(0.0, 0)
        f, n = t.track(*args)
        assert np.array_equal(f, g["flag"]) and t.stats()["fused_pass"] == 1


def test_bounded_waits_on_the_time_shard_path():
    """the same give-up on ctk_track_sharded_*: every rank reads it from the gathered headers and all repeat the call with one
    launch per filter pass"""
    from shard_inproc import sharded_threads
    g = golden_util.load("busy_s1")
    T = g["anom"].shape[0]
    want_n = len(np.unique(g["flag"])) - 1
    hs = [_native.Tracker(0) for _ in range(3)]
    try:
        hs[1].debug_set_spin(20.0, 2)                          # rank 1's chain head never publishes (its shard spans two workgroups)
        f, n, st = sharded_threads(hs, g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"],
                                   [0, 8, T - 8, T])
        assert np.array_equal(f, g["flag"]) and n == want_n
        assert all(s["off_fused_path_reason"] & 8 for s in st), st
    finally:
        for h in hs:
            h.close()


@pytest.mark.parametrize("shape", [(5, 1, 7), (5, 7, 1), (3, 2, 2), (4, 3, 64), (4, 3, 65), (2, 5, 128), (6, 4, 4), (9, 33, 3), (1, 1, 1),
                                   (7, 2, 4100)], ids=str)
def test_degenerate_grids(trk, oracle_lib, shape):
    """single rows / columns, widths below one store quad, exactly one / just over one mask word, > 64 words"""
    T, ny, nx = shape
    rng = np.random.default_rng(ny * 1000 + nx)
    a = (rng.random((T, ny, nx)) < 0.55).astype(np.float32)
    lat = np.linspace(60, -60, ny).astype(np.float32) if ny > 1 else np.array([10.0], dtype=np.float32)
    w = oracle_lib.row_weights(lat, np.float32(1.5), np.float32(2.0))
    thr = oracle_lib.prepare_thresholds(0.5, T)
    for two in (True, False):
        want, nw = oracle_lib.run_contrack(a, thr, ">=", w, 0.4, 2, two)
        got, ng = trk.track(a, thr, 0, w, 0.4, 2, two)
        assert np.array_equal(got, want) and ng == nw


def test_empty_slab(trk):
    f, n = trk.track(np.zeros((0, 5, 8), dtype=np.float32), np.zeros(0), 0, np.ones(5, dtype=np.float32), 0.5, 2, True)
    assert f.shape == (0, 5, 8) and n == -1                  # len(np.unique([])) - 1


def test_bad_arguments_raise(trk):
    a = np.zeros((2, 3, 4), dtype=np.float32)
    with pytest.raises(ValueError):
        trk.track(a, np.zeros(2), 7, np.ones(3, dtype=np.float32), 0.5, 2, True)              # cmp_op out of range
    with pytest.raises(ValueError):
        trk.track(a, np.zeros(2), 0, np.array([1.0, np.inf, 1.0], dtype=np.float32), 0.5, 2, True)   # non-finite weight
    with pytest.raises(ValueError):
        trk.track(a, np.zeros(3), 0, np.ones(3, dtype=np.float32), 0.5, 2, True)              # thr has the wrong length


def test_workspace_reuse_and_determinism(trk, oracle_lib):
    """same handle, different shapes back to back, then the first again: identical output."""
    g1, g2 = golden_util.load("syn2deg_s1"), golden_util.load("odd_17x64")
    outs = []
    for g in (g1, g2, g1, g2):
        f, _ = trk.track(g["anom"], g["thr"], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
        outs.append(f)
    assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[3])
    assert np.array_equal(outs[0], g1["flag"]) and np.array_equal(outs[1], g2["flag"])


def test_size_independent_properties(trk):
    """At a size the oracle is not run on (BASELINE config-1 grid, 400 steps):
    (a) the foreground of flag is a subset of the threshold mask;
    (b) every id lives at least `persistence` steps and n_tracked counts the distinct ids;
    (c) idempotence: tracking the output's own foreground with overlap -1 / persistence 1 reproduces the
        same partition with an order-preserving renumbering (the overlap filter removes nothing -- 0 would not do:
        the negative pole-row weights make fractions slightly negative -- and the 3-D
        components and the bbox-confined seam merges of the surviving pixels are unchanged)."""
    T, ny, nx = 400, 181, 360
    a = synth.smooth_field(T, ny, nx, seed=21)
    lat = np.linspace(90, -90, ny).astype(np.float32)
    w = (111 * np.float32(1.0) * 111 * np.float32(1.0) * np.cos(lat * np.pi / 180)).astype(np.float32)
    thr = np.full(T, np.float64(np.float32(160.0)))
    f, n = trk.track(a, thr, 0, w, 0.5, 5, True)
    assert ((f > 0) <= (a >= np.float32(160.0))).all()
    ids = np.unique(f)
    assert n == len(ids) - 1 and n > 10
    for i in ids[1:][:50]:
        ts = np.nonzero((f == i).any(axis=(1, 2)))[0]
        assert ts.max() - ts.min() + 1 >= 5
    f2, n2 = trk.track((f > 0).astype(np.float32), np.full(T, 0.5), 0, w, -1.0, 1, True)
    assert n2 == n
    assert np.array_equal(f2 > 0, f > 0)
    ids2 = np.unique(f2)
    lut = np.zeros(int(ids2.max()) + 1, dtype=np.int32)
    lut[ids2] = ids                      # order-preserving renumbering
    assert np.array_equal(lut[f2], f)


# ------------------------------------------------------------------------------------------------
# the product's time-sharded path with one PROCESS per rank: several processes share GPU 0 and talk through the shared-memory
# transport (RCCL refuses two ranks on one device; on a multi-GPU node the same entry runs over RCCL -- contrack_amd/dist.py)
# ------------------------------------------------------------------------------------------------
def _shard_worker(rank, world, key, name, q, backend="shm"):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_PORT=str(key), CTK_LAUNCH_PID=str(key),
                      CTK_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    from contrack_amd import dist as cdist
    g = golden_util.load(name)
    T = g["anom"].shape[0]
    t0, t1 = cdist.shard_bounds(T, world)[rank]
    st = cdist.ShardedTracker()
    try:
        flag, n = st.track(g["anom"][t0:t1], t0, T, g["thr"][t0:t1], _native.CMP_OPS[g["gorl"]], g["wrow"], g["overlap"], g["persistence"], g["twosided"])
        ok = bool(np.array_equal(flag, g["flag"][t0:t1])) and n == len(np.unique(g["flag"])) - 1
        q.put((rank, ok, n))
    finally:
        st.close()


@pytest.mark.parametrize("name,world", [("syn2deg_s0", 2), ("chain_a", 3), ("busy_s1", 4), ("noise", 2), ("T3", 3), ("syn1deg", 3), ("f64pole_blocky", 2)])
def test_time_sharded_processes_shm_transport(name, world):
    import multiprocessing as mp
    import os
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = 40000 + os.getpid() % 20000
    procs = [ctx.Process(target=_shard_worker, args=(r, world, key, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert len({n for _, _, n in res}) == 1


@pytest.mark.parametrize("name,world", [("syn2deg_s0", 2), ("chain_a", 3), ("busy_s1", 4), ("f64pole_blocky", 2), ("syn1deg", 8)])
def test_time_sharded_processes_rccl(name, world):
    """the same over RCCL, one process per GPU: runs where the node has at least `world` GPUs (the one-GPU box of the build
    skips it: RCCL refuses two ranks on one device) -- the first place ncclSend / ncclRecv / ncclAllGather of csrc/ctk_comm.hip
    meet more than one rank"""
    import multiprocessing as mp
    import os
    if _native.device_count() < world:
        pytest.skip("needs %d GPUs" % world)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = 20000 + os.getpid() % 20000
    procs = [ctx.Process(target=_shard_worker, args=(r, world, key, name, q, "rccl")) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=600) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(ok for _, ok, _ in res), res
    assert len({n for _, _, n in res}) == 1
