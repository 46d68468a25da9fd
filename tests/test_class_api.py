"""The drop-in class (contrack_amd/contrack.py) against the reference's own API tests
(/root/reference/tests/test_contrack.py:28-91), on the xarray stand-in tests/minixr.py when xarray is absent."""
import numpy as np
import pytest

import golden_util
import minixr
from contrack_amd.contrack import contrack, prepare_thresholds, row_weights

minixr.install_as_xarray()          # only when the real package is absent (it is, in this image)


def _dataset(name="refslab_fwd", dims=("time", "latitude", "longitude"), dtype=None):
    g = golden_util.load(name)
    a = g["anom"] if dtype is None else g["anom"].astype(dtype)
    order = [("time", "latitude", "longitude").index(d) for d in dims]
    ds = minixr.make_dataset(a.transpose(order), g["lat"], g["lon"], time_units="days since 2016-10-02", dims=("time", "latitude", "longitude"))
    if dims != ("time", "latitude", "longitude"):
        ds["anom"] = minixr.DataArray(a.transpose(order), dims, attrs={"units": "m", "long_name": "Z500 anomaly"})
    return ds, g


def test_init_empty():
    assert contrack().ds is None                                     # test_contrack.py:28-30


def test_read_xarray_and_properties():
    ds, _ = _dataset()
    c = contrack()
    c.read_xarray(ds)
    assert c.ds is ds
    assert len(c) == 1                                               # test_contrack.py:51-52
    assert c.ntime == 11                                             # :54-55
    assert c.dimensions == ['latitude', 'longitude', 'time']         # :57-58
    assert c.variables == ['anom']                                   # :60-61
    with pytest.raises(ValueError, match="already set"):
        c.read_xarray(ds)                                            # contrack.py:197-199
    with pytest.raises(ValueError, match="already set"):
        c.read("x.nc")
    assert contrack(ds=ds).ds is ds


def test_read_xarray_rejects_non_dataset():
    c = contrack()
    with pytest.raises(ValueError, match="ds has to be a xarray data set"):
        c.read_xarray(np.zeros(3))


def test_read_wrong_file():
    with pytest.raises(IOError) as e:
        contrack("tests/golden/does_not_exist.txt")
    assert e.value.args[0] == "Unkown fileformat. Known formats are netcdf."      # test_contrack.py:39-44


def test_set_up_manual_and_automatic():
    ds, _ = _dataset()
    c = contrack(ds=ds)
    c.set_up(time_name='time', longitude_name='longitude', latitude_name='latitude')
    assert (c._time_name, c._longitude_name, c._latitude_name) == ('time', 'longitude', 'latitude')
    c2 = contrack(ds=ds)
    c2.set_up()
    assert (c2._time_name, c2._longitude_name, c2._latitude_name) == ('time', 'longitude', 'latitude')
    assert c2._dlat.dtype == np.float32 and float(c2._dlat[0]) == 1.0 and float(c2._dlon[0]) == 1.0


def test_irregular_grid_raises_unless_forced():
    g = golden_util.load("cesm_like")
    ds = minixr.make_dataset(g["anom"], g["lat"], g["lon"])
    c = contrack(ds=ds)
    with pytest.raises(ValueError, match="No regular grid found for dimension latitude"):
        c.set_up()
    c.set_up(force=True)
    assert np.array_equal(np.asarray(c._dlat), g["dlat"])
    assert np.array_equal(row_weights(ds["latitude"].data, c._dlat, c._dlon), g["wrow"])


def test_bad_gorl_message():
    ds, _ = _dataset()
    c = contrack(ds=ds)
    with pytest.raises(ValueError) as e:
        c.run_contrack(variable='anom', threshold=150, gorl='=>', overlap=0.5, persistence=5)
    assert e.value.args[0] == ' Please select from [>, >=, <, >=] for gorl'       # contrack.py:658


def test_threshold_promotion_rules():
    # Python number -> cast to the float32 array dtype; float64 vector -> float64 compare (contrack.py:650, :665)
    t = prepare_thresholds(160.1, 3, np.float32)
    assert t.dtype == np.float64 and t[0] == float(np.float32(160.1)) and t.shape == (3,)
    v = np.array([1.1, 2.2, 3.3])
    assert np.array_equal(prepare_thresholds(v, 3, np.float32), v)
    assert prepare_thresholds(np.float32(1.1), 2, np.float32)[0] == float(np.float32(1.1))
    assert prepare_thresholds(160.1, 2, np.float64)[0] == 160.1
    assert prepare_thresholds(np.float64(1.1), 2, np.float32)[0] == 1.1


def test_greatcircle():
    d = contrack().greatcircle_dist(-0.27, 51.28, -73.46, 40.38)
    assert abs(d - 5555) < 30                                         # contrack.py:939-941 example


@pytest.mark.gpu
@pytest.mark.parametrize("name,dims", [("refslab_fwd", ("time", "latitude", "longitude")),
                                       ("refslab_two", ("latitude", "longitude", "time")),
                                       ("syn2deg_le", ("time", "latitude", "longitude"))])
def test_run_contrack_class(name, dims):
    """tests/test_contrack.py:83-91 through the drop-in class; ids equal the reference's."""
    ds, g = _dataset(name, dims)
    c = contrack()
    c.read_xarray(ds)
    thr = float(g["thr"][0])
    c.run_contrack(variable='anom', threshold=thr, gorl=g["gorl"], overlap=g["overlap"], persistence=g["persistence"],
                   twosided=g["twosided"])
    assert c.variables == ['anom', 'flag']
    flag = np.asarray(c.flag)                                          # __getattr__ delegation, contrack.py:110-113
    assert c['flag'].dims == dims
    order = [dims.index(d) for d in ("time", "latitude", "longitude")]
    assert np.array_equal(flag.transpose(order), g["flag"])
    assert c['flag'].attrs['units'] == 'flag' and 'threshold = ' in c['flag'].attrs['history']
    if name == "refslab_fwd":
        assert len(np.unique(c.flag)) - 1 == 3                        # test_contrack.py:91


@pytest.mark.gpu
@pytest.mark.parametrize("dims,chunk", [(("time", "latitude", "longitude"), 4), (("latitude", "longitude", "time"), 3), (("time", "latitude", "longitude"), 0)])
def test_run_contrack_class_streaming(dims, chunk):
    """chunk_steps: the variable is read slice by slice (isel) and streamed through the GPU; same flag variable"""
    ds, g = _dataset("refslab_two", dims)
    c = contrack(ds=ds)
    c.run_contrack(variable='anom', threshold=float(g["thr"][0]), gorl=g["gorl"], overlap=g["overlap"], persistence=g["persistence"],
                   twosided=g["twosided"], chunk_steps=chunk)
    assert c['flag'].dims == dims
    order = [dims.index(d) for d in ("time", "latitude", "longitude")]
    assert np.array_equal(np.asarray(c.flag).transpose(order), g["flag"])


@pytest.mark.gpu
def test_run_contrack_float64_and_dayofyear_threshold():
    g = golden_util.load("thr_vector")
    T = g["anom"].shape[0]
    time = (np.datetime64("2001-01-01") + np.arange(T)).astype("datetime64[ns]")
    ds = minixr.make_dataset(g["anom"].astype(np.float64), g["lat"], g["lon"], time=time)
    ds["time"].attrs = {}
    thr = minixr.DataArray(g["thr"], ("dayofyear",), coords={"dayofyear": minixr.DataArray(np.arange(1, T + 1), ("dayofyear",))})
    c = contrack(ds=ds)
    c.set_up(time_name="time", longitude_name="longitude", latitude_name="latitude")
    c.run_contrack(variable='anom', threshold=thr, gorl=g["gorl"], overlap=g["overlap"], persistence=g["persistence"], twosided=g["twosided"])
    assert np.array_equal(np.asarray(c.flag), g["flag"])


@pytest.mark.gpu
def test_run_lifecycle_known_answer():
    """tests/test_contrack.py:93-103: 3 flags, 28 rows on the reference's test slab (integer time axis);
    numeric columns cross-checked against a direct evaluation (tests/test_lifecycle.py holds the reference's values)."""
    ds, g = _dataset("refslab_fwd")
    ds["flag"] = minixr.DataArray(g["flag"], ("time", "latitude", "longitude"))
    c = contrack(ds=ds)
    df = c.run_lifecycle(flag="flag", variable="anom")
    assert list(df.columns) == ['Flag', 'Date', 'Longitude', 'Latitude', 'Intensity', 'Size']
    assert len(df.Flag.unique()) == 3 and len(df) == 28
    assert list(df.Flag) == sorted(df.Flag)
    w = g["wrow"].astype(np.float64)[:, None] * np.ones((1, 360))
    for _, r in df.iterrows():
        t = int(r.Date)                      # the stand-in's time coordinate is the integer day
        m = g["flag"][t] == r.Flag
        assert abs(r.Size - w[m].sum()) < 0.01 + 1e-9 * abs(w[m].sum())
        assert abs(r.Intensity - (w[m] * g["anom"][t][m]).sum() / w[m].sum()) < 0.006
        assert 0 <= r.Longitude < 360 and -90 <= r.Latitude <= 90
        ys, xs = np.nonzero(m)
        assert g["lat"][ys].min() - 1 <= r.Latitude <= g["lat"][ys].max() + 1


@pytest.mark.gpu
def test_flag_dtype_follows_scipy(monkeypatch):
    """int32 ids below 2^31 - 2 elements, int64 from there on -- scipy.ndimage.label's rule, hence the reference's (contrack.py:687,
    :751); the switch-over is exercised by lowering it"""
    import contrack_amd.contrack as mod
    g = golden_util.load("refslab_fwd")
    ds = minixr.make_dataset(g["anom"], g["lat"], g["lon"])
    c = mod.contrack(ds=ds)
    c.run_contrack("anom", threshold=float(g["thr"][0]), gorl=g["gorl"], overlap=float(g["overlap"]), persistence=int(g["persistence"]), twosided=bool(g["twosided"]))
    f32 = np.asarray(c["flag"].data)
    assert f32.dtype == np.int32 and np.array_equal(f32, g["flag"])
    monkeypatch.setattr(mod, "INT64_FLAG_FROM", f32.size)
    c.run_contrack("anom", threshold=float(g["thr"][0]), gorl=g["gorl"], overlap=float(g["overlap"]), persistence=int(g["persistence"]), twosided=bool(g["twosided"]))
    f64 = np.asarray(c["flag"].data)
    assert f64.dtype == np.int64 and np.array_equal(f64, g["flag"])
