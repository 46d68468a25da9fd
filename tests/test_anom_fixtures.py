"""Row N2 of SURVEY.md section 8(f) against the reference's OWN calc_clim / calc_anom outputs, when tests/golden/anom/*.npz exist
(tests/golden/make_anom_golden.py writes them on a machine with xarray; neither the build container nor the GPU box has it).
Without fixtures both tests skip and say why: parity with xarray is then pinned only by the pandas cross-check
(tests/test_anom_pandas.py).  Tolerance: the reference's float32 means run through xarray / bottleneck in float32 or float64
depending on its version, the port and the kernels sum in float64 and round once: 8 float32 ulp of the field's magnitude (6000)
for float32 slabs, 1e-9 for float64."""
import glob
import os

import numpy as np
import pytest

from oracle import anom_port

FIX = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "anom", "*.npz")))
UNPINNED = "parity unpinned: no tests/golden/anom/*.npz (written by tests/golden/make_anom_golden.py where xarray and the reference are installed)"


def _load(path):
    g = np.load(path)
    days = np.asarray(g["days"])
    dates = np.datetime64("2000-01-01") + days.astype("timedelta64[D]")
    doy = (dates - dates.astype("datetime64[Y]")).astype(int) + 1
    uniq, group = np.unique(doy, return_inverse=True)
    assert np.array_equal(uniq, np.asarray(g["clim_doy"]))
    return g, group.astype(np.int32), len(uniq)


def _tol(dtype):
    return 8 * np.finfo(np.float32).eps * 6000.0 if dtype == np.float32 else 1e-9


@pytest.mark.skipif(not FIX, reason=UNPINNED)
@pytest.mark.parametrize("path", FIX or ["-"])
def test_port_equals_reference_fixture(path):
    g, group, ng = _load(path)
    z = g["z"]
    clim = anom_port.calc_clim(z, group, ng, int(g["window"]))
    anom = anom_port.calc_anom(z, group, ng, int(g["window"]), int(g["smooth"]))
    np.testing.assert_allclose(clim, g["clim"], rtol=0, atol=_tol(z.dtype), equal_nan=True)
    np.testing.assert_allclose(anom, g["anom"], rtol=0, atol=_tol(z.dtype), equal_nan=True)


@pytest.mark.gpu
@pytest.mark.skipif(not FIX, reason=UNPINNED)
@pytest.mark.parametrize("path", FIX or ["-"])
def test_hip_equals_reference_fixture(path):
    from contrack_amd import _native
    g, group, ng = _load(path)
    z = g["z"]
    with _native.Tracker(0) as t:
        anom, clim = t.anomalies(z, group, ng, window=int(g["window"]), smooth=int(g["smooth"]), want_clim=True)
    np.testing.assert_allclose(clim, g["clim"], rtol=0, atol=_tol(z.dtype), equal_nan=True)
    np.testing.assert_allclose(anom, g["anom"], rtol=0, atol=_tol(z.dtype), equal_nan=True)
