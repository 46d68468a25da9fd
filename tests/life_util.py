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
