"""SURVEY.md section 8(f) rows N2 / N3 on the GPU (contrack_amd/csrc/ctk_anom.hip) against the numpy restatement
oracle/anom_port.py (PARITY UNPINNED: the reference's xarray calls cannot be run here -- see the port's header).
Tolerances: the kernels and the port both sum in float64 and round to the slab's dtype at the same places, but not in the
same order: 2 ulp of the dtype (float32: rtol 3e-7, atol scaled to the data)."""
import numpy as np
import pytest

import golden_util
import minixr
from contrack_amd import _native
from contrack_amd.contrack import contrack
from oracle import anom_port

pytestmark = pytest.mark.gpu
minixr.install_as_xarray()


@pytest.fixture(scope="module")
def trk():
    t = _native.Tracker(0)
    yield t
    t.close()


def _field(rng, T, ny, nx, dtype, nans):
    x = (50.0 * rng.standard_normal((T, ny, nx)) + 5500.0 + 30.0 * np.sin(np.arange(T) * 2 * np.pi / 365.0)[:, None, None]).astype(dtype)
    if nans:
        x[rng.random(x.shape) < 0.01] = np.nan
        x[:, 0, 0] = np.nan                      # a grid point without data
    return x


@pytest.mark.parametrize("case", [(400, 13, 20, np.float32, 1, 1, 0), (800, 9, 16, np.float32, 31, 2, 0), (800, 9, 16, np.float32, 4, 5, 1),
                                  (500, 7, 12, np.float64, 7, 3, 1), (366, 5, 8, np.float64, 1, 4, 0), (90, 6, 8, np.float32, 5, 1, 1),
                                  (1100, 181, 360, np.float32, 31, 2, 0), (380, 192, 288, np.float64, 5, 3, 1)],           # the 1 deg and CESM grids, three years
                         ids=str)
def test_anomalies_match_numpy_port(trk, case):
    T, ny, nx, dtype, window, smooth, nans = case
    rng = np.random.default_rng(T + window)
    x = _field(rng, T, ny, nx, dtype, nans)
    doy = (np.arange(T) + 17) % 365                       # day of year of a daily axis (0-based); every id present once T >= 365
    uniq, group = np.unique(doy, return_inverse=True)
    G = len(uniq)
    anom, clim = trk.anomalies(x, group, G, window=window, smooth=smooth, want_clim=True)
    want_c = anom_port.calc_clim(x, group, G, window)
    want_a = anom_port.calc_anom(x, group, G, window, smooth)
    eps = np.finfo(dtype).eps
    assert clim.dtype == dtype and anom.dtype == dtype
    assert np.array_equal(np.isnan(clim), np.isnan(want_c)) and np.array_equal(np.isnan(anom), np.isnan(want_a))
    np.testing.assert_allclose(clim, want_c.astype(dtype), rtol=3 * eps, atol=0)
    np.testing.assert_allclose(anom, want_a, rtol=0, atol=4 * eps * 6000.0)           # differences of ~5500-valued data
    # a climatology handed in (the `clim=` argument): same anomalies
    anom2, _ = trk.anomalies(x, group, G, window=window, smooth=smooth, clim=clim)
    assert np.array_equal(anom2, anom, equal_nan=True)


def test_percentile_threshold_matches_numpy(trk):
    rng = np.random.default_rng(3)
    for dtype in (np.float32, np.float64):
        x = (100.0 * rng.standard_normal((733, 31, 24))).astype(dtype)
        x[rng.random(x.shape) < 0.02] = np.nan
        x[:, 12, 5] = np.nan
        x[:, 13, :] = np.round(x[:, 13, :], -2)                      # many equal values: duplicates around the selected rank
        for q in (0.9, 0.5, 0.0, 1.0, 0.123):
            got = trk.percentile(x, 10, 20, q)
            want = anom_port.percentile_threshold(x, (10, 20), q)
            assert abs(got - want) <= 1e-12 * max(1.0, abs(want)), (dtype, q, got, want)


def _daily_dataset(T, ny, nx, seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    lat = np.linspace(90, -90, ny).astype(np.float32)
    lon = (np.arange(nx) * (360.0 / nx)).astype(np.float32)
    time = np.datetime64("1999-12-01") + np.arange(T).astype("timedelta64[D]")
    z = _field(rng, T, ny, nx, dtype, False)
    ds = minixr.make_dataset(z, lat, lon, time=time.astype("datetime64[ns]"), var="z", time_units="days since 1999-12-01")
    return ds, z


def test_class_calc_anom_then_run_contrack_from_hbm(trk):
    """calc_anom on the device; the slab stays resident and run_contrack uses it (no H2D) -- same flag as from the host array"""
    T, ny, nx = 800, 46, 72
    ds, z = _daily_dataset(T, ny, nx, seed=5)
    c = contrack(ds=ds)
    c.calc_anom('z', window=5, smooth=2)
    assert c.variables == ['z', 'anom']
    a = np.asarray(c['anom'].data)
    doy = np.asarray(c.ds['time'].dt.dayofyear)
    uniq, group = np.unique(doy, return_inverse=True)
    want = anom_port.calc_anom(z, group, len(uniq), 5, 2)
    np.testing.assert_allclose(a, want, rtol=0, atol=4 * np.finfo(np.float32).eps * 6000.0)
    assert c['anom'].attrs['long_name'].endswith(' Anomaly') and 'smoothing time steps = 2' in c['anom'].attrs['history']
    clim = c.calc_clim('z', window=5)
    assert tuple(clim.dims) == ('dayofyear', 'latitude', 'longitude') and np.asarray(clim.data).shape == (len(uniq), ny, nx)
    thr = c.percentile_threshold('anom', q=0.9, lat_bounds=(50, 80))
    lat = np.asarray(ds['latitude'].data)
    rows = np.nonzero((lat >= 50) & (lat <= 80))[0]
    assert abs(thr - anom_port.percentile_threshold(a, (rows[0], rows[-1] + 1), 0.9)) < 1e-9
    from contrack_amd.contrack import _tracker
    assert _tracker().resident_anom() == (T, ny, nx, False)
    c.run_contrack('anom', threshold=40.0, gorl='>=', overlap=0.5, persistence=3)
    from_hbm = np.asarray(c['flag'].data).copy()
    life_hbm = c.run_lifecycle(flag='flag', variable='anom')         # field = the resident slab: only the flags cross PCIe
    keep = c._anom_resident
    c._anom_resident = None                                          # force the host-array path
    c.run_contrack('anom', threshold=40.0, gorl='>=', overlap=0.5, persistence=3)
    assert np.array_equal(from_hbm, np.asarray(c['flag'].data)) and from_hbm.max() > 0
    life_host = c.run_lifecycle(flag='flag', variable='anom')
    assert len(life_hbm) > 10 and life_hbm.equals(life_host)
    c._anom_resident = keep
    # an edited slab must not be served from the stale device copy: the host copy is read-only while its twin lives in HBM ...
    c.calc_anom('z', window=5, smooth=2)
    with pytest.raises(ValueError, match="read-only"):
        c.ds['anom'].data[3, 4, 5] = 0.0
    # ... a single pixel edited after making it writeable again (invisible to any sampled check) is seen ...
    arr = c.ds['anom'].data
    if arr.base is not None:
        arr.base.flags.writeable = True                              # (the variable holds a transposed view of the library's array)
    arr.flags.writeable = True
    hot = np.unravel_index(np.argmax(arr[1:-1]), arr[1:-1].shape)
    arr[:, hot[1], hot[2]] = 0.0
    c.run_contrack('anom', threshold=40.0, gorl='>=', overlap=0.5, persistence=3)
    edited = np.asarray(c['flag'].data).copy()
    assert not np.array_equal(edited, from_hbm) and (edited[:, hot[1], hot[2]] == 0).all()
    # ... and so is a replaced variable
    c.calc_anom('z', window=5, smooth=2)
    c.ds['anom'] = (c['anom'].dims, np.zeros_like(np.asarray(c['anom'].data)), dict(c['anom'].attrs))
    c.run_contrack('anom', threshold=40.0, gorl='>=', overlap=0.5, persistence=3)
    assert np.asarray(c['flag'].data).max() == 0


def test_two_instances_do_not_share_a_resident_slab(trk):
    """A.calc_anom, B.calc_anom on a same-shaped dataset, A.run_contrack: the slab resident in HBM is B's by then -- A must run on
    ITS anomalies (the resident slab is tied to the call that produced it: ctk_resident_anom_generation), B may use the slab"""
    from contrack_amd.contrack import row_weights
    T, ny, nx = 400, 31, 60
    a, b = (contrack(ds=_daily_dataset(T, ny, nx, seed=s)[0]) for s in (3, 4))
    a.calc_anom("z", window=3, smooth=1)
    anom_a = np.array(a.ds["anom"].data)
    b.calc_anom("z", window=3, smooth=1)                       # same shape, same handle: B's slab replaces A's in HBM
    anom_b = np.array(b.ds["anom"].data)
    assert anom_a.shape == anom_b.shape and not np.array_equal(anom_a, anom_b)
    a.run_contrack("anom", threshold=40.0, gorl=">=", overlap=0.5, persistence=2)
    b.run_contrack("anom", threshold=40.0, gorl=">=", overlap=0.5, persistence=2)
    for c, anom in ((a, anom_a), (b, anom_b)):                 # references: the same call on plain host copies
        w = row_weights(np.asarray(c.ds["latitude"].data), c._dlat, c._dlon)
        want, _ = trk.track(np.ascontiguousarray(anom), np.full(T, np.float64(np.float32(40.0))), 0, w, 0.5, 2, True)
        assert want.max() > 0 and np.array_equal(np.asarray(c.ds["flag"].data), want)
    assert not np.array_equal(np.asarray(a.ds["flag"].data), np.asarray(b.ds["flag"].data))


def test_calc_anom_with_a_climatology_that_lacks_a_group(trk):
    """a `clim=` without one of the data's day-of-year values: a ValueError that says so (not a bare KeyError); the climatology
    keeps the variable's name"""
    T, ny, nx = 400, 9, 16
    ds, _ = _daily_dataset(T, ny, nx, seed=1)
    c = contrack(ds=ds)
    c.set_up()
    clim = c.calc_clim("z")
    assert getattr(clim, "name", None) == "z"
    n = np.asarray(clim.data).shape[0]
    short = type(clim)(np.asarray(clim.data)[:n - 5], dims=clim.dims, coords={"dayofyear": np.asarray(clim["dayofyear"].data)[:n - 5],
                       "latitude": np.asarray(ds["latitude"].data), "longitude": np.asarray(ds["longitude"].data)})
    with pytest.raises(ValueError, match="dayofyear"):
        c.calc_anom("z", clim=short)
