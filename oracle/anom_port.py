"""numpy restatement of contrack.calc_clim / calc_anom (contrack/contrack.py:458-581) and of the README's percentile
threshold (README.rst:150-151), for checking the HIP kernels of contrack_amd/csrc/ctk_anom.hip.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED against the reference itself: it evaluates these steps with xarray (groupby().mean(),
rolling(center=True).mean(), fillna, groupby arithmetic, quantile), and xarray cannot be installed in the build container, so the
reference cannot be run on them here.  Two things stand in: (1) tests/test_anom_pandas.py -- the same steps a second time with
pandas' groupby / rolling / fillna (whose semantics xarray documents as its own), equal to this module for odd and even windows,
NaNs and a supplied climatology; (2) tests/golden/make_anom_golden.py, which writes fixtures from the reference's own calc_clim /
calc_anom wherever xarray is installed -- tests/test_anom_fixtures.py consumes them when present and pins the row then.
What is restated is the documented behaviour of the xarray calls:

  calc_clim  (:482-489)  clim_raw[g] = mean over the timesteps of group g (NaNs skipped: xarray's mean has skipna=True for floats);
                         clim[g] = mean of clim_raw over the centred window of `window` groups, NaN where the window leaves the
                         axis (rolling's min_periods defaults to the window), then every NaN replaced by the mean of the LAST
                         `window` groups of clim_raw (fillna(clim[-window:].mean())) -- at both ends of the axis.
  calc_anom  (:566-570)  anom_raw[t] = x[t] - clim[group(t)];  anom[t] = mean of anom_raw over the centred window of `smooth`
                         timesteps, NaN where the window leaves the axis or holds a NaN.
  centred window of w    xarray computes the trailing window and shifts it by (-w // 2) + 1: it covers [i - w // 2, i + (w - 1) // 2]
                         (one more element on the left for even w).
  percentile threshold   anom.sel(latitude=band).quantile([q], dim='time').mean(): per grid point the q-quantile over time
                         (numpy's default linear interpolation, NaNs skipped), then the mean over the band's grid points.

Sums are taken in float64; results are cast to the dtype of the input where xarray would keep it (float32 in, float32 out).
"""
import numpy as np


def centred_window(i, w):
    return i - w // 2, i + (w - 1) // 2               # inclusive bounds


def rolling_mean_centred(a, w, axis=0):
    """mean over the centred window of w along `axis`; NaN where the window leaves the axis or contains a NaN"""
    a = np.moveaxis(np.asarray(a, dtype=np.float64), axis, 0)
    n = a.shape[0]
    out = np.full(a.shape, np.nan)
    for i in range(n):
        lo, hi = centred_window(i, w)
        if lo < 0 or hi >= n:
            continue
        out[i] = a[lo:hi + 1].sum(axis=0) / w          # (a NaN in the window makes the sum NaN)
    return np.moveaxis(out, 0, axis)


def calc_clim(x, group, ngroups, window=1):
    """x (T, ny, nx); group[t] in [0, ngroups).  Returns clim (ngroups, ny, nx) float64."""
    x = np.asarray(x)
    raw = np.full((ngroups,) + x.shape[1:], np.nan)
    for g in range(ngroups):
        sel = x[np.asarray(group) == g].astype(np.float64)
        if len(sel):
            cnt = np.sum(~np.isnan(sel), axis=0)
            with np.errstate(invalid="ignore", divide="ignore"):
                raw[g] = np.where(cnt > 0, np.nansum(sel, axis=0) / np.maximum(cnt, 1), np.nan)
    raw = raw.astype(x.dtype).astype(np.float64) if x.dtype == np.float32 else raw
    clim = rolling_mean_centred(raw, window, axis=0)
    tail = raw[-window:]
    cnt = np.sum(~np.isnan(tail), axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        fill = np.where(cnt > 0, np.nansum(tail, axis=0) / np.maximum(cnt, 1), np.nan)
    return np.where(np.isnan(clim), fill[None], clim)


def calc_anom(x, group, ngroups, window=1, smooth=1, clim=None):
    x = np.asarray(x)
    if clim is None:
        clim = calc_clim(x, group, ngroups, window)
    if x.dtype == np.float32:
        clim = np.asarray(clim, dtype=np.float64).astype(np.float32).astype(np.float64)
    raw = x.astype(np.float64) - np.asarray(clim, dtype=np.float64)[np.asarray(group)]
    if x.dtype == np.float32:
        raw = raw.astype(np.float32).astype(np.float64)
    return rolling_mean_centred(raw, smooth, axis=0).astype(x.dtype)


def percentile_threshold(anom, rows, q):
    """mean over the grid points of rows [rows[0], rows[1]) of the q-quantile over time (linear interpolation, NaNs skipped)"""
    band = np.asarray(anom)[:, rows[0]:rows[1], :].astype(np.float64)
    with np.errstate(invalid="ignore"):
        qv = np.nanquantile(band, q, axis=0)
    return float(np.nanmean(qv))                              # (xarray's mean skips the grid points without data)
