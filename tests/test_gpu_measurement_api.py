"""The measurement entries of include/contrack_hip_debug.h that bench.py and tools/ rely on (round 6): they must not change a result."""
import numpy as np
import pytest

from contrack_amd import _native, synth

pytestmark = pytest.mark.gpu


def _slab(T=40, ny=181, nx=360, seed=9):
    a = synth.smooth_field(T, ny, nx, seed=seed)
    lat, _ = synth.grid(ny, nx)
    w = np.array(111 * np.float32(180 / (ny - 1)) * 111 * np.float32(360 / nx) * np.cos(lat * np.pi / 180)).astype(np.float32)
    return a, w


def test_write_kernel_variants_write_the_same_flags():
    """ctk_debug_time_relabel: k_relabel_v5 with and without its SGPR limit, in the three chunk -> XCD orders, on the finished tables of a
    pass -- every launch has to reproduce the pass's own flag slab."""
    a, w = _slab()
    T, ny, nx = a.shape
    with _native.Tracker(0) as trk:
        d_in, d_out = trk.malloc(a.nbytes), trk.malloc(a.nbytes)
        trk.h2d(d_in, a)
        thr = np.full(T, 160.0)
        n = trk.track_dev(d_in, T, ny, nx, thr, 0, w, 0.5, 5, True, d_out)
        ref = tuple(trk.checksum_i32(d_out, T * ny * nx))
        assert n > 0
        for variant in (0, 1):
            for xcd in (-1, 0, 1, 16):
                trk.memset(d_out, 0x5a, a.nbytes)
                ms = trk.time_relabel(d_out, 5, variant, xcd, reps=2)
                assert (ms > 0).all()
                assert tuple(trk.checksum_i32(d_out, T * ny * nx)) == ref, (variant, xcd)
        with pytest.raises((ValueError, _native.ContrackHipError)):
            trk.time_relabel(d_out, 5, 7, -1, reps=1)
        trk.free(d_in)
        trk.free(d_out)


def test_plain_streams_run_in_every_mode():
    with _native.Tracker(0) as trk:
        nbytes = 64 << 20
        p = trk.malloc(nbytes)
        trk.memset(p, 1, nbytes)
        for mode in (0, 1, 2):
            assert trk.stream_ceiling(p, nbytes, mode, reps=2) > 0
        assert trk.checksum_i32(p, nbytes // 4)[1] == 0        # the store streams wrote zeros over the whole buffer
        trk.free(p)
