"""GPU-free host logic: ctk_resolve (contrack_amd/csrc/ctk_resolve.cpp) on tables built on the CPU,
checked against the reference's golden outputs; single shard and split into several shards."""
import numpy as np
import pytest

import cpu_tables
import golden_util
from contrack_amd import _native


def _mask_of(g, oracle_lib):
    return oracle_lib.threshold_mask(g["anom"], g["thr"], g["gorl"]).astype(bool)


def _run(g, mask, splits):
    wlo, whi, wshift = _native.weights_to_limbs(g["wrow"])
    T = mask.shape[0]
    bounds = [0] + list(splits) + [T]
    blobs, labs, prev = [], [], None
    for s in range(len(bounds) - 1):
        a, b = bounds[s], bounds[s + 1]
        tb = cpu_tables.build_tables(mask[a:b], wlo, whi, prev_lab=prev)
        blobs.append(cpu_tables.pack_blob(tb, wshift, has_prev=prev is not None))
        labs.extend(tb["labs"])
        if b > a:
            prev = tb["labs"][-1]
    res = _native.resolve(blobs, g["overlap"], g["twosided"])
    comp_label, ops = res.arrays()
    info = res.info()
    flag = cpu_tables.apply_result(labs, comp_label, ops, g["persistence"])
    return flag, info


@pytest.mark.parametrize("name", golden_util.case_names())
def test_resolve_single_shard_matches_golden(oracle_lib, name):
    g = golden_util.load(name)
    mask = _mask_of(g, oracle_lib)
    flag, info = _run(g, mask, [])
    assert np.array_equal(flag, g["flag"])
    assert info["n_ambiguous"] == 0


@pytest.mark.parametrize("name", ["syn2deg_s0", "busy_s0", "busy_s2", "noise", "refslab_two", "T3", "odd_9x65"])
@pytest.mark.parametrize("nsplit", [2, 3, 5])
def test_resolve_sharded_matches_golden(oracle_lib, name, nsplit):
    g = golden_util.load(name)
    mask = _mask_of(g, oracle_lib)
    T = mask.shape[0]
    splits = sorted(set(int(round(T * k / nsplit)) for k in range(1, nsplit)))
    flag, _ = _run(g, mask, splits)
    assert np.array_equal(flag, g["flag"])


def test_resolve_rejects_garbage():
    with pytest.raises(ValueError):
        _native.resolve([b"\0" * 128], 0.5, True)


def test_weight_limbs_exact_and_range():
    w = np.array([-5.3856801e-04, 215.03082, 12321.0, 0.0, 1e-30], dtype=np.float32)
    with pytest.raises(ValueError):
        _native.weights_to_limbs(w)                       # 1e-30 .. 1e4 spans more than 62 bits
    w = w[:4]
    lo, hi, sh = _native.weights_to_limbs(w)
    for i in range(4):
        assert (int(lo[i]) + int(hi[i]) * 2 ** 31) / 2.0 ** sh == float(w[i])
    with pytest.raises(ValueError):
        _native.weights_to_limbs(np.array([1.0, np.inf], dtype=np.float32))
