"""run_lifecycle (next row N1): oracle port and host finishing step against the reference's own frames
(tests/golden/life), then the HIP reductions (ctk_lifecycle_*) against both."""
import numpy as np
import pytest

import life_util
import minixr
from contrack_amd.contrack import contrack, lifecycle_frame, row_weights
from oracle import lifecycle_port

minixr.install_as_xarray()
CASES = life_util.case_names()


def test_fixtures_present():
    assert {"refslab", "smooth0", "smooth1", "smooth2", "ring", "float64", "othervar"} <= set(CASES)


@pytest.mark.parametrize("name", CASES)
def test_oracle_port_matches_reference(name):
    g = life_util.load(name)
    got = lifecycle_port.run_lifecycle(g["flag"], g["variable"], g["lat"], g["lon"], g["wrow"], life_util.dates_of(g["time"]))
    assert got == g["frame"]                       # same call sequence as the reference: identical, digit for digit


@pytest.mark.parametrize("name", CASES)
def test_host_finish_matches_reference(name):
    """lifecycle_frame (divisions, int(), coordinate look-ups, rounding) on numpy-built rows"""
    g = life_util.load(name)
    rows = life_util.numpy_rows(g["flag"], g["variable"], g["wrow"])
    got = lifecycle_frame(rows, g["lat"], g["lon"], life_util.dates_of(g["time"]))
    _compare(got, g["frame"])


def test_fixtures_cover_the_seam_roll():
    rolled = 0
    for name in CASES:
        g = life_util.load(name)
        rows = life_util.numpy_rows(g["flag"], g["variable"], g["wrow"])
        rolled += int((rows["shift"] > 0).sum())
    assert rolled >= 20
    ring = life_util.numpy_rows(life_util.load("ring")["flag"], life_util.load("ring")["variable"], life_util.load("ring")["wrow"])
    assert (ring["shift"] == 1).any()              # every column occupied: np.argmax of equal gaps -> roll by cols[1]


def test_finish_errors():
    from contrack_amd._native import LIFE_ROW
    lat, lon = np.linspace(90, -90, 5), np.arange(4.0)
    r = np.zeros(1, dtype=LIFE_ROW)
    r["label"], r["shift"], r["area"], r["swv"] = 3, -1, 2.0, 0.0
    with pytest.raises(ValueError, match="NaN"):
        lifecycle_frame(r, lat, lon, ["d0"])       # int(nan) in the reference
    r["shift"] = -2
    with pytest.raises(ValueError, match="argmax"):
        lifecycle_frame(r, lat, lon, ["d0"])
    r["shift"], r["swv"], r["swvy"], r["swvx"] = -1, 1.0, -0.5, -1.5     # negative centre: int() truncates, Python index wraps
    out = lifecycle_frame(r, lat, lon, ["d0"])
    assert out[0][2] == int(lon[-1]) and out[0][3] == int(lat[0])
    r["swvx"] = 99.0
    with pytest.raises(IndexError):
        lifecycle_frame(r, lat, lon, ["d0"])


def _compare(got, want, tol=0.0101):
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a[:4] == b[:4], (a, b)
        assert abs(a[4] - b[4]) <= tol and abs(a[5] - b[5]) <= tol * max(1.0, abs(b[5]) * 1e-9), (a, b)


# ---------------------------------------------------------------------------------------------------------------
# HIP
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tracker():
    from contrack_amd import _native
    with _native.Tracker(0) as t:
        yield t


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_rows_and_frame_match_reference(tracker, name):
    g = life_util.load(name)
    rows = tracker.lifecycle(g["flag"], g["variable"], g["wrow"])
    want = life_util.numpy_rows(g["flag"], g["variable"], g["wrow"])
    assert np.array_equal(rows["t"], want["t"]) and np.array_equal(rows["label"], want["label"])
    assert np.array_equal(rows["shift"], want["shift"])
    assert np.array_equal(rows["area"], want["area"])                 # exact: integer limbs, rounded once
    for k in ("swv", "swvy", "swvx"):
        assert np.allclose(rows[k], want[k], rtol=1e-12, atol=1e-9), k
    _compare(lifecycle_frame(rows, g["lat"], g["lon"], life_util.dates_of(g["time"])), g["frame"])


@pytest.mark.gpu
def test_class_run_lifecycle_known_answer():
    """tests/test_contrack.py:93-103 through the drop-in class: 3 flags, 28 rows, and the reference's values"""
    g = life_util.load("refslab")
    ds = minixr.make_dataset(g["field"], g["lat"], g["lon"], time=g["time"])
    ds["time"].attrs = {}
    ds["flag"] = minixr.DataArray(g["flag"].astype(np.int64), ("time", "latitude", "longitude"))
    c = contrack(ds=ds)
    c.set_up(time_name="time", longitude_name="longitude", latitude_name="latitude")
    df = c.run_lifecycle(flag="flag", variable="anom")
    assert list(df.columns) == ['Flag', 'Date', 'Longitude', 'Latitude', 'Intensity', 'Size']
    assert len(df.Flag.unique()) == 3 and len(df) == 28
    _compare([tuple(r) for r in df.itertuples(index=False)], g["frame"])


@pytest.mark.gpu
@pytest.mark.parametrize("f64", [False, True])
def test_hip_lifecycle_on_tracked_slab(tracker, f64):
    """track a synthetic slab on the GPU, then life cycle of its flags against the scipy port"""
    from contrack_amd import synth
    T, ny, nx = 40, 91, 180
    anom = synth.smooth_field(T, ny, nx, seed=5)
    if f64:
        anom = anom.astype(np.float64) + 1e-7
    lat, lon = synth.grid(ny, nx)
    wrow = row_weights(lat, 2.0, 2.0)
    flag, n = tracker.track(anom, np.full(T, 100.0), 0, wrow, 0.4, 3, True, f64=f64)
    assert n > 3
    rows = tracker.lifecycle(flag, anom, wrow)
    dates = ["%03d" % t for t in range(T)]
    got = lifecycle_frame(rows, lat, lon, dates)
    want = lifecycle_port.run_lifecycle(flag, anom, lat, lon, wrow, dates)
    assert (rows["shift"] > 0).any()
    _compare(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,thr,pers", [((6, 721, 1440), 160.0, 2), ((24, 181, 360), 160.0, 3), ((8, 192, 288), 150.0, 2)])
def test_hip_lifecycle_on_baseline_grids(tracker, shape, thr, pers):
    """the grids of BASELINE.json (0.25 deg: six 256-column strips x twelve 64-row bands per plane; 1 deg; CESM): frame identical to
    the scipy port's in every column, with the rows on rounding boundaries re-evaluated on the device"""
    from contrack_amd import synth
    T, ny, nx = shape
    anom = synth.smooth_field(T, ny, nx, seed=9)
    lat, lon = synth.grid(ny, nx)
    wrow = row_weights(lat, 180.0 / (ny - 1), 360.0 / nx)
    flag, n = tracker.track(anom, np.full(T, thr), 0, wrow, 0.5, pers, True)
    assert n > 2
    rows = tracker.lifecycle(flag, anom, wrow)
    dates = ["%03d" % t for t in range(T)]
    got = lifecycle_frame(rows, lat, lon, dates, tracker)
    assert (rows["shift"] > 0).any()                                   # contours across the seam
    assert got == lifecycle_port.run_lifecycle(flag, anom, lat, lon, wrow, dates)


@pytest.mark.gpu
def test_hip_lifecycle_device_resident_and_empty(tracker):
    from contrack_amd import synth
    T, ny, nx = 6, 46, 72
    rng = np.random.default_rng(0)
    flag = np.zeros((T, ny, nx), dtype=np.int32)
    field = rng.random((T, ny, nx), dtype=np.float32) + 1
    wrow = row_weights(synth.grid(ny, nx)[0], 4.0, 5.0)
    assert len(tracker.lifecycle(flag, field, wrow)) == 0
    assert len(tracker.lifecycle(flag[:0], field[:0], wrow)) == 0
    flag[2, 10:14, 70:] = 7
    flag[2, 10:12, :3] = 7
    flag[3, 5, 5] = -4                                            # any non-zero id counts (labels != 0, contrack.py:866)
    fd, vd = tracker.malloc(flag.nbytes), tracker.malloc(field.nbytes)
    try:
        tracker.h2d(fd, flag)
        tracker.h2d(vd, field)
        rows = tracker.lifecycle_dev(fd, vd, T, ny, nx, wrow)
    finally:
        tracker.free(fd)
        tracker.free(vd)
    want = life_util.numpy_rows(flag, field, wrow)
    assert [tuple(r)[:3] for r in rows] == [tuple(r)[:3] for r in want] == [(3, -4, -1), (2, 7, 70)]
    assert np.array_equal(rows["area"], want["area"]) and np.allclose(rows["swvx"], want["swvx"], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(40))
def test_hip_lifecycle_random_frames(tracker, i):
    """random label planes (thin contours whose centre of mass is an integer up to rounding included): the frame equals
    the scipy port's in every column, digit for digit"""
    flag, field, lat, lon, wrow, dates = life_util.random_life_case(i)
    rows = tracker.lifecycle(flag, field, wrow)
    got = lifecycle_frame(rows, lat, lon, dates, tracker)
    assert got == lifecycle_port.run_lifecycle(flag, field, lat, lon, wrow, dates)
    # ... and with EVERY row re-evaluated in the reference's summation orders on the device: the sums themselves are the
    # reference's, bit for bit (np.sum pairwise for area / intensity, np.bincount sequential for the centre of mass)
    ex = tracker.lifecycle_exact(np.arange(len(rows)))
    want = life_util.numpy_exact_rows(flag, field, wrow, rows)
    for k in ("area", "swv", "s", "sy", "sx"):
        assert np.array_equal(ex[k], want[k]), k


@pytest.mark.gpu
def test_hip_lifecycle_limits(tracker):
    ny, nx = 40, 64
    wrow = row_weights(np.linspace(60, 21, ny, dtype=np.float32), 1.0, 1.0)
    flag = np.zeros((1, ny, nx), dtype=np.int32)
    flag[0, ::2, ::2] = np.arange(1, 20 * 32 + 1).reshape(20, 32)          # 640 ids in one time step: more than the LDS tables hold,
    rows = tracker.lifecycle(flag, np.ones(flag.shape, np.float32), wrow)   # processed in passes over residue classes of the ids
    want = life_util.numpy_rows(flag, np.ones(flag.shape, np.float32), wrow)
    assert len(rows) == 640 and np.array_equal(rows["label"], want["label"]) and np.array_equal(rows["area"], want["area"])
    big = np.zeros((2, ny, nx), dtype=np.int32)
    big[1] = np.arange(1, ny * nx + 1).reshape(ny, nx)                      # 2560 ids, every one its own pixel; 40 of them touch both seam columns? no: one column each
    big[1, :, -1] = big[1, :, 0]                                            # ... now 40 ids cross the seam (more than the 32 column bit sets)
    rows = tracker.lifecycle(big, np.ones(big.shape, np.float32), wrow)
    want = life_util.numpy_rows(big, np.ones(big.shape, np.float32), wrow)
    assert len(rows) == len(want) and np.array_equal(rows["label"], want["label"]) and np.array_equal(rows["shift"], want["shift"])
    assert np.array_equal(rows["area"], want["area"]) and np.allclose(rows["swvx"], want["swvx"], rtol=1e-12)
    flag[0, ::2, ::2] = np.arange(1, 20 * 32 + 1).reshape(20, 32) % 500 + 1 # 500 ids: fits in one pass
    rows = tracker.lifecycle(flag, np.ones(flag.shape, np.float32), wrow)
    assert len(rows) == 500
    want = life_util.numpy_rows(flag, np.ones(flag.shape, np.float32), wrow)
    assert np.array_equal(rows["area"], want["area"]) and np.allclose(rows["swvy"], want["swvy"], rtol=1e-12)
    with pytest.raises(ValueError):
        tracker.lifecycle(flag, np.ones((1, ny, nx + 1), np.float32), wrow)
