#!/usr/bin/env python3
"""Golden vectors for run_lifecycle: runs the UNMODIFIED reference class (/root/reference/contrack/contrack.py,
imported under tests/minixr.py) -- run_contrack to get `flag`, then run_lifecycle(flag, variable) -- on seeded inputs
and stores inputs + the reference's DataFrame columns under tests/golden/life/.  Build container only.

    python tests/golden/make_life_golden.py

Each .npz: field (float32 as int16 / q_scale, or float64 raw; 'field_ref' names refslab_input.npz), flag (int32),
lat, lon, wrow (float32), time (datetime64[h] as int64 hours since 1970), and the reference's columns
Flag / Date / Longitude / Latitude / Intensity / Size.
"""
import logging
import os
import sys
import warnings

import numpy as np
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import minixr  # noqa: E402
import refimport  # noqa: E402

OUT = os.path.join(HERE, "life")


def smooth_field(T, ny, nx, seed, sigma=(1.5, 3.0, 5.0), amp=260.0):
    rng = np.random.default_rng(seed)
    a = ndimage.gaussian_filter(rng.standard_normal((T, ny, nx)), sigma, mode=("nearest", "nearest", "wrap"))
    return a / np.abs(a).max() * amp


def grid(ny, nx):
    lat = np.linspace(90, -90, ny).astype(np.float32)
    lon = (np.arange(nx) * (360.0 / nx)).astype(np.float32)
    return lat, lon


def reference_run(field, lat, lon, time, threshold, gorl, overlap, persistence, twosided, variable_field=None):
    cls = refimport.load()
    # the reference's set_up cannot take datetime64 time steps under pandas >= 2 (TimedeltaIndex.astype('timedelta64[h]'),
    # contrack.py:338): set up on the integer time axis the other goldens use, then give the data set its dates for
    # run_lifecycle's strftime (contrack.py:862)
    ds = minixr.make_dataset(field, lat, lon)
    if variable_field is not None:
        ds["other"] = minixr.DataArray(variable_field, ("time", "latitude", "longitude"))
    c = cls()
    c.read_xarray(ds)
    logging.disable(logging.CRITICAL)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c.set_up(time_name="time", longitude_name="longitude", latitude_name="latitude")
            c.run_contrack(variable="anom", threshold=threshold, gorl=gorl, overlap=overlap, persistence=persistence, twosided=twosided)
            ds["time"].data = np.asarray(time)
            df = c.run_lifecycle(flag="flag", variable="anom" if variable_field is None else "other")
    finally:
        logging.disable(logging.NOTSET)
    wrow = np.array(111 * c._dlat * 111 * c._dlon * np.cos(np.asarray(lat) * np.pi / 180)).astype(np.float32)
    return np.asarray(c.ds["flag"].data).astype(np.int32), df, wrow


def save(name, field, flag, lat, lon, wrow, time, df, variable=None, field_ref=None):
    d = dict(flag=flag, lat=lat, lon=lon, wrow=wrow, time=time.astype("datetime64[h]").astype(np.int64),
             Flag=np.asarray(df.Flag, dtype=np.int64), Date=np.asarray(df.Date, dtype=str), Longitude=np.asarray(df.Longitude, dtype=np.int64),
             Latitude=np.asarray(df.Latitude, dtype=np.int64), Intensity=np.asarray(df.Intensity, dtype=np.float64),
             Size=np.asarray(df.Size, dtype=np.float64))
    if field_ref is not None:
        d["field_ref"] = field_ref
    elif field.dtype == np.float64:
        d["field64"] = field
    else:
        q = np.round(field.astype(np.float64) * 8).astype(np.int16)
        assert np.array_equal((q / 8.0).astype(np.float32), field)
        d["field_q"] = q
    if variable is not None:
        v = np.round(variable.astype(np.float64) * 8).astype(np.int32)
        assert np.array_equal((v / 8.0).astype(np.float32), variable)
        d["variable_q"] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("%-14s rows %4d  ids %3d  seam-rolled rows: see test" % (name, len(df), len(set(df.Flag))))


def q8(a):
    return (np.round(a * 8) / 8).astype(np.float32)


def main():
    os.makedirs(OUT, exist_ok=True)
    hours = lambda T, step=6, start="2001-12-30T00": (np.datetime64(start, "h") + np.arange(T) * step).astype("datetime64[ns]")

    # the reference's own test slab and call (tests/test_contrack.py:83-103)
    raw = np.load(os.path.join(HERE, "refslab_input.npz"))["anom"].astype(np.float32)
    lat, lon = np.linspace(90, -90, 181, dtype=np.float32), np.arange(360, dtype=np.float32)
    t = hours(raw.shape[0], 24, "2016-10-02T00")
    flag, df, wrow = reference_run(raw, lat, lon, t, 150, ">=", 0.5, 5, False)
    assert len(df) == 28 and len(set(df.Flag)) == 3                      # the reference's own known answer
    save("refslab", raw, flag, lat, lon, wrow, t, df, field_ref="refslab_input.npz")

    # smooth synthetic fields: several contours, some across the seam
    for k, (thr, gorl, ov, pers) in enumerate([(60, ">=", 0.3, 2), (-70, "<=", 0.2, 2), (45, ">", 0.0, 1)]):
        lat, lon = grid(46, 72)
        f = q8(smooth_field(16, 46, 72, 100 + k))
        t = hours(16)
        flag, df, wrow = reference_run(f, lat, lon, t, thr, gorl, ov, pers, True)
        save("smooth%d" % k, f, flag, lat, lon, wrow, t, df)

    # a zonal ring (every column occupied -> roll by 1, contrack.py:883) next to ordinary contours
    lat, lon = grid(37, 60)
    f = smooth_field(10, 37, 60, 7, amp=120.0)
    f[:, 8:11, :] += 200.0
    f = q8(f)
    t = hours(10, 12)
    flag, df, wrow = reference_run(f, lat, lon, t, 150, ">=", 0.1, 2, True)
    save("ring", f, flag, lat, lon, wrow, t, df)

    # float64 field, and a second positive variable for intensity / centre of mass
    lat, lon = grid(31, 48)
    f = smooth_field(8, 31, 48, 21).astype(np.float64)
    t = hours(8, 3)
    flag, df, wrow = reference_run(f, lat, lon, t, 55.5, ">=", 0.25, 2, True)
    save("float64", f, flag, lat, lon, wrow, t, df)
    other = q8(smooth_field(8, 31, 48, 22, amp=40.0) + 500.0)
    flag2, df2, _ = reference_run(q8(f), lat, lon, t, 55.5, ">=", 0.25, 2, True, variable_field=other)
    save("othervar", q8(f), flag2, lat, lon, wrow, t, df2, variable=other)


if __name__ == "__main__":
    main()
