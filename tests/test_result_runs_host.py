"""The decoder of the run-table result transfer (host-array entries) on its own, without a device: tables built
in numpy from a dense int32 slab -> ctk_expand_runs_host -> the slab again."""
import numpy as np
import pytest

from contrack_amd import _native


def _tables(flag, values_of_runs=None):
    """bit mask, first run of every row, runs per time step and one value per run (the value of the run's pixels) of a slab whose
    foreground (!= 0) runs are constant -- the form the device pass ends with"""
    T, ny, nx = flag.shape
    W = (nx + 63) // 64
    fg = flag != 0
    bits = np.zeros((T, ny, W * 64), dtype=bool)
    bits[:, :, :nx] = fg
    mask = np.packbits(bits.reshape(T, ny, W, 64), axis=-1, bitorder="little").view(np.uint64).reshape(T, ny, W)
    start = fg & ~np.concatenate([np.zeros((T, ny, 1), bool), fg[:, :, :-1]], axis=2)
    per_row = start.sum(axis=2)
    rowstart = (np.cumsum(per_row, axis=1) - per_row).astype(np.uint32)
    run_base = np.concatenate([[0], np.cumsum(per_row.sum(axis=1))]).astype(np.uint32)
    run_val = flag[start].astype(np.int32)                      # raster order = run order
    return mask, rowstart, run_base, run_val


def _blobs(rng, T, ny, nx, density):
    """constant-valued horizontal runs (ids from the mask's maximal runs, so that a run never changes value)"""
    fg = rng.random((T, ny, nx)) < density
    if density > 0.5:
        fg |= rng.random((T, ny, 1)) < 0.3                      # some entirely foreground rows
    start = fg & ~np.concatenate([np.zeros((T, ny, 1), bool), fg[:, :, :-1]], axis=2)
    run_id = np.cumsum(start.reshape(-1)).reshape(T, ny, nx)
    vals = rng.integers(0, 50, size=int(run_id.max()) + 1).astype(np.int32)      # 0: a run filtered out by persistence
    return np.where(fg, vals[run_id], 0).astype(np.int32), fg


@pytest.mark.parametrize("shape", [(3, 5, 64), (2, 7, 65), (4, 9, 130), (1, 3, 1), (2, 4, 63), (3, 6, 360), (2, 3, 1440), (5, 2, 200)])
@pytest.mark.parametrize("density", [0.05, 0.4, 0.9])
def test_decoder_reproduces_the_slab(shape, density):
    rng = np.random.default_rng(hash((shape, density)) & 0xffff)
    T, ny, nx = shape
    flag, fg = _blobs(rng, T, ny, nx, density)
    mask, rowstart, run_base, run_val = _tables(np.where(fg, np.where(flag == 0, -7, flag), 0))
    run_val = np.where(run_val == -7, 0, run_val).astype(np.int32)                # filtered runs: foreground in the mask, value 0
    got, zero, cx = _native.expand_runs_host(mask, rowstart, run_base, run_val, nx)
    assert np.array_equal(got, flag)
    assert zero == bool((flag == 0).any()) and not cx


def test_all_foreground_and_all_background():
    T, ny, nx = 2, 3, 128
    full = np.full((T, ny, nx), 9, np.int32)
    got, zero, cx = _native.expand_runs_host(*_tables(full), nx)
    assert np.array_equal(got, full) and not zero and not cx
    empty = np.zeros((T, ny, nx), np.int32)
    got, zero, cx = _native.expand_runs_host(*_tables(empty), nx)
    assert np.array_equal(got, empty) and zero and not cx


def test_negative_run_values_are_reported_not_decoded():
    flag = np.zeros((1, 2, 70), np.int32)
    flag[0, 0, 3:9] = 4
    flag[0, 1, 60:70] = 5
    mask, rowstart, run_base, run_val = _tables(flag)
    run_val = run_val.copy()
    run_val[1] = -5                                              # a complex component: the caller sends the block to the write kernel
    got, zero, cx = _native.expand_runs_host(mask, rowstart, run_base, run_val, 70)
    assert cx and np.array_equal(got[0, 0], flag[0, 0]) and (got[0, 1] == 0).all()


def test_bad_shapes_are_rejected():
    with pytest.raises(ValueError):
        _native.expand_runs_host(np.zeros((1, 2, 2), np.uint64), np.zeros((1, 2), np.uint32), np.zeros(2, np.uint32), np.zeros(0, np.int32), 200)
