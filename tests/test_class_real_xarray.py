"""The drop-in class on the REAL xarray package.

xarray is not installed in the build container nor on the GPU box (and cannot be: no network), so every test here is
SKIPPED in this image; the module exists for a maintainer who has xarray: `python -m pytest tests/test_class_real_xarray.py`
repeats the class-level checks of tests/test_class_api.py (which run on the stand-in tests/minixr.py) on real
xr.Dataset / xr.DataArray objects, incl. datetime64 time coordinates and a dayofyear threshold array.  The xarray
members the class touches are listed in INTEGRATION.md section 5a.
"""
import sys

import numpy as np
import pytest

import golden_util
from contrack_amd.contrack import contrack


def _real_xarray():
    mod = sys.modules.get("xarray")
    if mod is not None:                                   # tests/minixr.py may have registered itself under that name
        return mod if getattr(mod, "__file__", None) else None
    try:
        import xarray
        return xarray
    except ImportError:
        return None


xr = _real_xarray()
pytestmark = pytest.mark.skipif(xr is None, reason="the real xarray package is not installed")

DIMS = ("time", "latitude", "longitude")


def _dataset(name, dims=DIMS, dtype=None):
    g = golden_util.load(name)
    a = g["anom"] if dtype is None else g["anom"].astype(dtype)
    T = a.shape[0]
    time = (np.datetime64("2016-10-02") + np.arange(T)).astype("datetime64[ns]")
    order = [DIMS.index(d) for d in dims]
    ds = xr.Dataset({"anom": (dims, a.transpose(order), {"units": "m", "long_name": "Z500 anomaly"})},
                    coords={"time": time, "latitude": ("latitude", g["lat"], {"units": "degrees_north"}),
                            "longitude": ("longitude", g["lon"], {"units": "degrees_east"})})
    return ds, g


def test_properties_and_set_up():
    ds, _ = _dataset("refslab_fwd")
    c = contrack()
    c.read_xarray(ds)
    assert c.ntime == 11 and c.variables == ['anom'] and sorted(c.dimensions) == ['latitude', 'longitude', 'time']
    c.set_up()
    assert (c._time_name, c._longitude_name, c._latitude_name) == ('time', 'longitude', 'latitude')
    assert float(c._dlat[0]) == 1.0 and float(c._dlon[0]) == 1.0
    with pytest.raises(ValueError, match="ds has to be a xarray data set"):
        contrack().read_xarray(np.zeros(3))


@pytest.mark.gpu
@pytest.mark.parametrize("name,dims", [("refslab_fwd", DIMS), ("refslab_two", ("latitude", "longitude", "time"))])
def test_run_contrack(name, dims):
    ds, g = _dataset(name, dims)
    c = contrack(ds=ds)
    c.run_contrack(variable='anom', threshold=float(g["thr"][0]), gorl=g["gorl"], overlap=g["overlap"],
                   persistence=g["persistence"], twosided=g["twosided"])
    assert isinstance(c['flag'], xr.DataArray) and c['flag'].dims == dims
    assert np.array_equal(c['flag'].transpose(*DIMS).data, g["flag"])
    assert c['flag'].attrs['units'] == 'flag'


@pytest.mark.gpu
def test_dayofyear_threshold_array():
    g = golden_util.load("thr_vector")
    T = g["anom"].shape[0]
    time = (np.datetime64("2001-01-01") + np.arange(T)).astype("datetime64[ns]")
    ds = xr.Dataset({"anom": (DIMS, g["anom"].astype(np.float64))}, coords={"time": time, "latitude": g["lat"], "longitude": g["lon"]})
    thr = xr.DataArray(g["thr"], dims=("dayofyear",), coords={"dayofyear": np.arange(1, T + 1)})
    c = contrack(ds=ds)
    c.set_up(time_name="time", longitude_name="longitude", latitude_name="latitude")
    c.run_contrack(variable='anom', threshold=thr, gorl=g["gorl"], overlap=g["overlap"], persistence=g["persistence"], twosided=g["twosided"])
    assert np.array_equal(c['flag'].data, g["flag"])


@pytest.mark.gpu
def test_calc_anom_against_xarray_itself():
    """the device calc_clim / calc_anom against the reference's own xarray expressions (contrack.py:482-489, :566-570)"""
    ds, g = _dataset("refslab_fwd", dtype=np.float32)
    c = contrack(ds=ds.rename({"anom": "z"}))
    c.ds["z"].attrs.update(units="m", long_name="Z500")
    c.set_up()
    c.calc_anom(variable="z", window=3, smooth=2, groupby="dayofyear")
    z = ds["anom"]
    clim = z.groupby("time.dayofyear").mean("time")
    clim = clim.rolling(dayofyear=3, center=True).mean().fillna(clim[-3:].mean(dim="dayofyear"))
    want = (z.groupby("time.dayofyear") - clim).rolling(time=2, center=True).mean()
    got = c["anom"].transpose(*DIMS).data
    assert np.array_equal(np.isnan(got), np.isnan(want.data))
    assert np.allclose(got, want.data, rtol=1e-6, atol=1e-4, equal_nan=True)


@pytest.mark.gpu
def test_run_lifecycle_known_answer():
    ds, g = _dataset("refslab_fwd")
    ds["flag"] = (DIMS, g["flag"])
    c = contrack(ds=ds)
    df = c.run_lifecycle(flag="flag", variable="anom")
    assert list(df.columns) == ['Flag', 'Date', 'Longitude', 'Latitude', 'Intensity', 'Size']
    assert len(df.Flag.unique()) == 3 and len(df) == 28                 # tests/test_contrack.py:93-103
    assert df.Date.iloc[0].startswith("2016")                           # dt.strftime('%Y%m%d_%H'), contrack.py:861
