"""ctypes front end of oracle/contrack_oracle.c (pixel-level restatement of contrack.py:646-796)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None

_OPS = {">=": 0, "ge": 0, "<=": 1, "le": 1, ">": 2, "gt": 2, "<": 3, "lt": 3}


def build(force=False):
    src = os.path.join(_HERE, "contrack_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        p = C.c_void_p
        L.orc_run_contrack.argtypes = [p, C.c_int64, C.c_int, C.c_int, p, C.c_int, p, C.c_double,
                                       C.c_int, C.c_int, p, p, p]
        L.orc_run_contrack.restype = C.c_int
        L.orc_threshold.argtypes = [p, C.c_int64, C.c_int, C.c_int, p, C.c_int, p]
        L.orc_threshold.restype = C.c_int
        L.orc_label.argtypes = [p, C.c_int64, C.c_int, C.c_int, C.c_int, p]
        L.orc_label.restype = C.c_int64
        L.orc_np_sum.argtypes = [p, C.c_int64]
        L.orc_np_sum.restype = C.c_double
        _lib = L
    return _lib


def row_weights(lat, dlat, dlon):
    """Row weights exactly as contrack/contrack.py:703-704 evaluates them (restated):
    cos(lat*pi/180) in the dtype of `lat`, times 111*dlat*111*dlon left to right, cast to float32."""
    lat = np.asarray(lat)
    weight_lat = np.cos(lat * np.pi / 180)
    return np.array((111 * dlat * 111 * dlon * weight_lat)).astype(np.float32).reshape(-1)


def prepare_thresholds(threshold, T, data_dtype=np.float32):
    """Per-timestep double thresholds such that `(double)x <op> thr[t]` reproduces the reference's
    compare (contrack.py:665): a Python number is a weak scalar and is cast to the array dtype;
    a numpy float64 scalar/array promotes the compare to float64."""
    if isinstance(threshold, (int, float)) and not isinstance(threshold, np.generic):
        thr = np.full(T, np.asarray(threshold, dtype=data_dtype).astype(np.float64))
    else:
        arr = np.asarray(threshold)
        if arr.dtype == np.float32 or arr.dtype.kind in "iu":
            arr = arr.astype(data_dtype)
        thr = np.broadcast_to(arr.astype(np.float64), (T,)).copy()
    return np.ascontiguousarray(thr, dtype=np.float64)


def run_contrack(anom, thr, gorl, wrow, overlap, persistence, twosided=True, return_stage=False):
    """anom: float32 (T,ny,nx) C-contiguous; thr: float64 (T,); wrow: float32 (ny,)."""
    L = lib()
    anom = np.ascontiguousarray(anom, dtype=np.float32)
    T, ny, nx = anom.shape
    thr = np.ascontiguousarray(thr, dtype=np.float64)
    wrow = np.ascontiguousarray(wrow, dtype=np.float32)
    assert thr.shape == (T,) and wrow.shape == (ny,)
    if gorl not in _OPS:
        raise ValueError(' Please select from [>, >=, <, >=] for gorl')
    flag = np.empty((T, ny, nx), dtype=np.int32)
    stage = np.empty((T, ny, nx), dtype=np.int32) if return_stage else None
    ntr = C.c_int64(0)
    rc = L.orc_run_contrack(anom.ctypes.data, T, ny, nx, thr.ctypes.data, _OPS[gorl], wrow.ctypes.data,
                            float(overlap), int(persistence), int(bool(twosided)), flag.ctypes.data,
                            C.addressof(ntr), stage.ctypes.data if return_stage else None)
    if rc != 0:
        raise RuntimeError("oracle failed rc=%d" % rc)
    if return_stage:
        return flag, int(ntr.value), stage
    return flag, int(ntr.value)


def threshold_mask(anom, thr, gorl):
    L = lib()
    anom = np.ascontiguousarray(anom, dtype=np.float32)
    T, ny, nx = anom.shape
    thr = np.ascontiguousarray(thr, dtype=np.float64)
    m = np.empty((T, ny, nx), dtype=np.uint8)
    rc = L.orc_threshold(anom.ctypes.data, T, ny, nx, thr.ctypes.data, _OPS[gorl], m.ctypes.data)
    assert rc == 0
    return m


def label(mask, temporal):
    L = lib()
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    T, ny, nx = mask.shape
    lab = np.empty((T, ny, nx), dtype=np.int32)
    n = L.orc_label(mask.ctypes.data, T, ny, nx, int(temporal), lab.ctypes.data)
    return lab, int(n)


def np_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().orc_np_sum(a.ctypes.data, a.size)
