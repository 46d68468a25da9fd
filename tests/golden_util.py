"""Loader for the golden fixtures written by tests/golden/make_golden.py."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names():
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")) if not p.endswith("refslab_input.npz"))


def _levels(z, shape):
    vin = np.broadcast_to(np.asarray(z["v_in"], dtype=np.float32).reshape(-1, 1, 1), shape)
    vout = np.broadcast_to(np.asarray(z["v_out"], dtype=np.float32).reshape(-1, 1, 1), shape)
    return vin, vout


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    if "anom" in z:
        a = np.array(z["anom"], dtype=np.float32)
    elif "anom_ref" in z:
        a = np.array(np.load(os.path.join(GOLDEN_DIR, str(z["anom_ref"])))["anom"], dtype=np.float32)
    else:
        shape = tuple(int(v) for v in z["shape"])
        n = int(np.prod(shape))
        m = np.unpackbits(z["mask_bits"])[:n].reshape(shape).astype(bool)
        vin, vout = _levels(z, shape)
        a = np.where(m, vin, vout).astype(np.float32)
        if "nan_bits" in z:
            a[np.unpackbits(z["nan_bits"])[:n].reshape(shape).astype(bool)] = np.nan
    flag = z["flag_palette"][z["flag_index"].astype(np.int64)].astype(np.int32)
    return dict(name=name, anom=np.ascontiguousarray(a), lat=z["lat"], lon=z["lon"], dlat=z["dlat"], dlon=z["dlon"],
                wrow=np.array(z["wrow"], dtype=np.float32), thr=np.array(z["thr"], dtype=np.float64),
                gorl=str(z["gorl"]), overlap=float(z["overlap"]), persistence=int(z["persistence"]),
                twosided=bool(z["twosided"]), flag=flag)
