"""Loader for tests/golden/life/*.npz (written by tests/golden/make_life_golden.py from the reference's own
run_lifecycle) and a numpy builder of ctk_life_row records used to test the host-side finishing step without a GPU."""
import glob
import os

import numpy as np

LIFE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "life")


def case_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(LIFE_DIR, "*.npz")))


def load(name):
    z = np.load(os.path.join(LIFE_DIR, name + ".npz"))
    if "field_ref" in z:
        field = np.load(os.path.join(os.path.dirname(LIFE_DIR), str(z["field_ref"])))["anom"].astype(np.float32)
    elif "field64" in z:
        field = np.array(z["field64"], dtype=np.float64)
    else:
        field = (z["field_q"] / 8.0).astype(np.float32)
    variable = (z["variable_q"] / 8.0).astype(np.float32) if "variable_q" in z else field
    time = z["time"].astype("datetime64[h]").astype("datetime64[ns]")
    frame = list(zip((int(v) for v in z["Flag"]), (str(v) for v in z["Date"]), (int(v) for v in z["Longitude"]),
                     (int(v) for v in z["Latitude"]), (float(v) for v in z["Intensity"]), (float(v) for v in z["Size"])))
    return dict(name=name, field=field, variable=variable, flag=np.array(z["flag"], dtype=np.int32), lat=z["lat"], lon=z["lon"],
                wrow=np.array(z["wrow"], dtype=np.float32), time=time, frame=frame)


def dates_of(time):
    import pandas as pd
    return [pd.Timestamp(v).strftime('%Y%m%d_%H') for v in time]


def numpy_rows(flags, field, wrow):
    """What ctk_lifecycle_* returns, evaluated with numpy (float64 sums in raster order)."""
    from contrack_amd._native import LIFE_ROW
    T, ny, nx = flags.shape
    w = np.asarray(wrow, dtype=np.float32).astype(np.float64)[:, None] * np.ones((1, nx))
    yy, xx = np.mgrid[0:ny, 0:nx]
    out = []
    for t in range(T):
        for ident in np.unique(flags[t]):
            if ident == 0:
                continue
            m = flags[t] == ident
            shift = -1
            if m[:, 0].any() and m[:, -1].any():
                cols = np.unique(np.nonzero(m)[1])
                shift = int(cols[np.argmax(np.diff(cols)) + 1]) if len(cols) > 1 else -2
            xr = (xx - shift) % nx if shift > 0 else xx
            wv = field[t].astype(np.float64) * w
            out.append((t, int(ident), shift, 0, w[m].sum(), wv[m].sum(), (wv * yy)[m].sum(), (wv * xr)[m].sum()))
    rows = np.array(out, dtype=LIFE_ROW) if out else np.empty(0, dtype=LIFE_ROW)
    return rows[np.lexsort((rows["t"], rows["label"]))]


def random_life_case(i):
    """random label planes (blobs, some joined across the seam, ids not contiguous, sometimes negative) and a positive
    field: (flag, field, lat, lon, wrow, dates)"""
    from scipy import ndimage
    from contrack_amd.contrack import row_weights
    rng = np.random.default_rng(90000 + i)
    T = int(rng.integers(1, 7)); ny = int(rng.integers(3, 40)); nx = int(rng.choice([4, 7, 16, 33, 64, 65, 100, 130]))
    # labelled blobs: threshold a smooth-ish random field, label with wrap-unaware scipy, then join ids across the seam at random
    f = ndimage.uniform_filter(rng.standard_normal((T, ny, nx)), size=(1, 3, 5), mode=("nearest", "nearest", "wrap"))
    flag = np.zeros((T, ny, nx), np.int32)
    for t in range(T):
        lab, n = ndimage.label(f[t] > 0.15)
        perm = rng.permutation(np.arange(1, n + 1)) * int(rng.choice([1, 1, 7])) if n else np.array([], int)
        flag[t] = np.where(lab > 0, np.concatenate([[0], perm])[lab], 0)
        for y in range(ny):                                  # merge across the seam sometimes
            if flag[t, y, 0] and flag[t, y, -1] and rng.random() < 0.7:
                flag[t][flag[t] == flag[t, y, -1]] = flag[t, y, 0]
    if rng.random() < 0.2:
        flag[flag == flag.max()] = -5                            # a negative id
    f64 = bool(rng.integers(0, 2))
    field = (rng.random((T, ny, nx)) * 50 + 100).astype(np.float64 if f64 else np.float32)
    lat = np.linspace(90, -90, ny).astype(np.float32); lon = (np.arange(nx) * (360.0 / nx)).astype(np.float32)
    wrow = row_weights(lat, 180.0 / (ny - 1), 360.0 / nx)
    dates = ["%02d" % t for t in range(T)]
    return flag, field, lat, lon, wrow, dates


def numpy_exact_rows(flags, field, wrow, rows):
    """what ctk_lifecycle_exact returns for `rows`, evaluated with the reference's own calls: np.sum for the area and the
    intensity numerator (contrack.py:874-875), np.bincount (what ndimage.center_of_mass sums with, :886 / :892) on the rolled plane"""
    from contrack_amd._native import LIFE_EXACT
    T, ny, nx = flags.shape
    wgrid = np.ones((ny, nx)) * np.asarray(wrow, dtype=np.float32)[:, None]
    yy, xx = np.mgrid[0:ny, 0:nx]
    out = np.zeros(len(rows), dtype=LIFE_EXACT)
    for i, r in enumerate(rows):
        plane, values = flags[r["t"]], field[r["t"]]
        m = plane == r["label"]
        sh = int(r["shift"]) if r["shift"] > 0 else 0
        pr, vr = np.roll(plane, -sh, axis=1), np.roll(values, -sh, axis=1)
        inp = vr * wgrid
        sel = (pr == r["label"]).ravel().astype(np.intp)
        out[i] = (np.sum(wgrid[m]), np.sum(wgrid[m] * values[m]), np.bincount(sel, weights=inp.ravel())[1],
                  np.bincount(sel, weights=(inp * yy.astype(float)).ravel())[1], np.bincount(sel, weights=(inp * xx.astype(float)).ravel())[1])
    return out
