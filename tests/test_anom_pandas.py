"""calc_clim / calc_anom a second time, independently, with pandas -- groupby().mean(), rolling(window, center=True).mean(), fillna:
the very operations the reference spells out with xarray (contrack/contrack.py:482-489, :566-570), whose rolling and groupby
semantics xarray documents as pandas' -- against the numpy restatement oracle/anom_port.py, which is what the HIP kernels of
csrc/ctk_anom.hip are checked with.  xarray itself is not installable in the build container; tests/golden/make_anom_golden.py writes
fixtures from the reference's own calc_clim / calc_anom on a machine that has it (tests/test_anom_fixtures.py consumes them), until
then this is the pin: two implementations by different means that must agree for odd and even windows, NaNs and a supplied `clim`."""
import numpy as np
import pandas as pd
import pytest

from oracle import anom_port


def _pandas_clim(x, group, ngroups, window):
    T = x.shape[0]
    flat = pd.DataFrame(x.reshape(T, -1).astype(np.float64))
    raw = flat.groupby(np.asarray(group)).mean().reindex(range(ngroups))                 # NaNs skipped, like xarray's mean
    if x.dtype == np.float32:
        raw = raw.astype(np.float32).astype(np.float64)                                   # (xarray keeps float32 data float32)
    clim = raw.rolling(window, center=True).mean()
    clim = clim.fillna(raw.iloc[-window:].mean())                                         # clim[-window:].mean(dim=groupby), :488
    return clim.to_numpy().reshape((ngroups,) + x.shape[1:])


def _pandas_anom(x, group, ngroups, window, smooth, clim=None):
    T = x.shape[0]
    if clim is None:
        clim = _pandas_clim(x, group, ngroups, window)
    if x.dtype == np.float32:
        clim = clim.astype(np.float32).astype(np.float64)
    raw = pd.DataFrame(x.reshape(T, -1).astype(np.float64) - clim.reshape(ngroups, -1)[np.asarray(group)])
    if x.dtype == np.float32:
        raw = raw.astype(np.float32).astype(np.float64)
    return raw.rolling(smooth, center=True).mean().to_numpy().reshape(x.shape).astype(x.dtype)


def _case(seed, T, ngroups, dtype, nans):
    rng = np.random.default_rng(seed)
    x = (5500 + 80 * rng.standard_normal((T, 5, 7))).astype(dtype)
    if nans:
        x[rng.random(x.shape) < 0.02] = np.nan
        x[:, 2, 3] = np.nan                                     # a grid point without data
    group = np.arange(T) % ngroups
    return x, group


@pytest.mark.parametrize("window", [1, 2, 3, 4, 5, 8, 31])
@pytest.mark.parametrize("dtype,nans", [(np.float64, False), (np.float64, True), (np.float32, False), (np.float32, True)])
def test_clim_port_equals_pandas(window, dtype, nans):
    x, g = _case(window, 400, 73, dtype, nans)
    got = anom_port.calc_clim(x, g, 73, window)
    want = _pandas_clim(x, g, 73, window)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-9, equal_nan=True)


@pytest.mark.parametrize("window,smooth", [(1, 1), (3, 2), (4, 3), (5, 4), (2, 5), (31, 2)])
@pytest.mark.parametrize("dtype,nans", [(np.float64, False), (np.float64, True), (np.float32, False), (np.float32, True)])
def test_anom_port_equals_pandas(window, smooth, dtype, nans):
    x, g = _case(100 + window + smooth, 300, 61, dtype, nans)
    got = anom_port.calc_anom(x, g, 61, window, smooth)
    want = _pandas_anom(x, g, 61, window, smooth)
    assert got.dtype == want.dtype and np.array_equal(np.isnan(got), np.isnan(want))
    # float64: both sum in float64, in different orders (the port: slice sums, pandas: an online window) -> a few ulp of the sums;
    # float32 results additionally round once to float32
    tol = 1e-9 if dtype == np.float64 else 4 * np.finfo(np.float32).eps * 6000.0
    np.testing.assert_allclose(got, want, rtol=0, atol=tol, equal_nan=True)


@pytest.mark.parametrize("smooth", [1, 2, 3])
def test_anom_with_a_supplied_climatology(smooth):
    x, g = _case(7, 200, 50, np.float64, False)
    clim = np.random.default_rng(1).standard_normal((50, 5, 7)) * 10 + 5500
    got = anom_port.calc_anom(x, g, 50, 1, smooth, clim=clim)
    want = _pandas_anom(x, g, 50, 1, smooth, clim=clim)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9, equal_nan=True)


def test_even_window_alignment_is_pandas():
    """the one convention that documentation alone leaves open: an even centred window has its extra element on the LEFT"""
    a = pd.Series(np.arange(8, dtype=np.float64))
    r4 = a.rolling(4, center=True).mean().to_numpy()
    mine = anom_port.rolling_mean_centred(a.to_numpy(), 4)
    assert np.array_equal(np.isnan(r4), np.isnan(mine)) and np.allclose(r4[2:7], mine[2:7]) and np.isnan(mine[0]) and np.isnan(mine[1]) and np.isnan(mine[7])
