"""contrack_amd -- MI355X (gfx950) implementation of ConTrack's run_contrack hot path.

`contrack` is the drop-in class (same constructor / set_up / calc_anom / run_contrack signatures and the
same 'flag' output variable as steidani/ConTrack's contrack.contrack); `track_numpy` is the array-level
entry underneath it.  Compute goes through hand-written HIP kernels behind a ctypes C ABI
(include/contrack_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"


def __getattr__(name):
    if name in ("contrack", "track_numpy", "row_weights", "prepare_thresholds"):
        from . import contrack as _m
        return getattr(_m, name)
    raise AttributeError(name)
