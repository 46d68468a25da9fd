#!/usr/bin/env python3
"""Writes tests/golden/anom/*.npz from the REFERENCE's own calc_clim / calc_anom (contrack/contrack.py:458-581) -- run it on a
machine that has xarray (the build container and the GPU box have not, and cannot install it):

    PYTHONPATH=/path/to/ConTrack python tests/golden/make_anom_golden.py

Every fixture holds the input slab, the time axis (days since 2000-01-01), the parameters and the reference's outputs
(`clim`, `anom`).  With the fixtures present, tests/test_anom_fixtures.py checks oracle/anom_port.py against them on the CPU and
the HIP kernels (ctk_anom_*) on the GPU, and the row "N2" of SURVEY.md section 8(f) is pinned; without them both tests skip with the
reason "parity unpinned".  The fixtures are data (inputs and the reference's outputs), a few hundred KB each.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "anom")

CASES = [  # name, T, ny, nx, dtype, window, smooth, nan_fraction, seed
    ("d1_w1_s1_f32", 800, 12, 16, "float32", 1, 1, 0.0, 0),
    ("d1_w5_s2_f32", 800, 12, 16, "float32", 5, 2, 0.0, 1),
    ("d1_w4_s3_f64", 800, 12, 16, "float64", 4, 3, 0.0, 2),
    ("d1_w31_s8_f32", 1500, 8, 12, "float32", 31, 8, 0.0, 3),
    ("d1_w5_s2_nan_f32", 800, 12, 16, "float32", 5, 2, 0.02, 4),
    ("d1_w2_s1_nan_f64", 800, 12, 16, "float64", 2, 1, 0.02, 5),
]


def main():
    import xarray as xr                                   # noqa: F401 -- the point of this script
    from contrack import contrack
    os.makedirs(OUT, exist_ok=True)
    for name, T, ny, nx, dtype, window, smooth, nanf, seed in CASES:
        rng = np.random.default_rng(seed)
        z = (5500 + 90 * rng.standard_normal((T, ny, nx))).astype(dtype)
        if nanf:
            z[rng.random(z.shape) < nanf] = np.nan
        time = np.datetime64("2000-01-01") + np.arange(T).astype("timedelta64[D]")
        lat = np.linspace(80, -80, ny).astype(np.float32)
        lon = (np.arange(nx) * (360.0 / nx)).astype(np.float32)
        ds = xr.Dataset({"z": (("time", "latitude", "longitude"), z, {"units": "gpm", "long_name": "Geopotential Height"})},
                        coords={"time": time, "latitude": ("latitude", lat, {"units": "degrees_north"}),
                                "longitude": ("longitude", lon, {"units": "degrees_east"})})
        c = contrack()
        c.read_xarray(ds)
        c.set_up()
        clim = c.calc_clim("z", window=window, groupby="dayofyear")
        c.calc_anom("z", window=window, smooth=smooth, groupby="dayofyear")
        np.savez_compressed(os.path.join(OUT, name + ".npz"), z=z, days=np.arange(T), window=window, smooth=smooth,
                            clim=np.asarray(clim.transpose("dayofyear", "latitude", "longitude").data),
                            clim_doy=np.asarray(clim["dayofyear"].data), anom=np.asarray(c.ds["anom"].data),
                            xarray_version=xr.__version__, numpy_version=np.__version__)
        print("wrote", name)


if __name__ == "__main__":
    sys.exit(main())
