"""The numpy restatement of calc_clim / calc_anom / the percentile threshold (oracle/anom_port.py) on hand-checkable inputs.
(It is the checker of the HIP kernels in tests/test_gpu_anom.py; parity with xarray itself is unpinned -- see its header.)"""
import numpy as np

from oracle import anom_port


def test_centred_window_convention():
    # xarray: trailing window shifted by (-w // 2) + 1 -> [i - w // 2, i + (w - 1) // 2]
    assert anom_port.centred_window(5, 1) == (5, 5)
    assert anom_port.centred_window(5, 3) == (4, 6)
    assert anom_port.centred_window(5, 4) == (3, 6)
    a = np.arange(6, dtype=np.float64)
    r = anom_port.rolling_mean_centred(a, 3)
    assert np.isnan(r[0]) and np.isnan(r[5]) and np.allclose(r[1:5], [1, 2, 3, 4])
    r = anom_port.rolling_mean_centred(a, 2)
    assert np.isnan(r[0]) and np.allclose(r[1:], [0.5, 1.5, 2.5, 3.5, 4.5])


def test_clim_fill_and_anomaly():
    # 3 "years" of 4 groups; one grid point
    x = np.array([1, 2, 3, 4, 3, 4, 5, 6, 5, 6, 7, 8], dtype=np.float64).reshape(12, 1, 1)
    g = np.tile(np.arange(4), 3)
    raw = np.array([3, 4, 5, 6], dtype=np.float64)
    c = anom_port.calc_clim(x, g, 4, window=3)[:, 0, 0]
    # centred means exist for groups 1, 2; the ends are filled with the mean of the LAST three groups of the raw climatology
    assert np.allclose(c, [raw[1:].mean(), 4.0, 5.0, raw[1:].mean()])
    a = anom_port.calc_anom(x, g, 4, window=1, smooth=1)[:, 0, 0]
    assert np.allclose(a, x[:, 0, 0] - raw[g])
    a2 = anom_port.calc_anom(x, g, 4, window=1, smooth=2)[:, 0, 0]
    assert np.isnan(a2[0]) and np.allclose(a2[1:], (a[:-1] + a[1:]) / 2)


def test_percentile_threshold():
    x = np.arange(10, dtype=np.float64).reshape(10, 1, 1) * np.ones((1, 3, 2))
    x[:, 1, :] *= 2
    assert np.isclose(anom_port.percentile_threshold(x, (0, 2), 0.9), (8.1 + 16.2) / 2)
