#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference
(/root/reference/contrack/contrack.py, imported under tests/minixr.py because xarray is not
installable here) on seeded inputs.  Build container only; the fixtures it writes are data
(inputs + the reference's outputs), committed so that the GPU box -- which has no /root/reference
-- can check both the CPU oracle and the HIP path against the reference's own results.

    python tests/golden/make_golden.py [case ...] # rewrites tests/golden/*.npz (or only the named cases)

Each .npz holds: anom (float32, or anom_q int16 + q_scale), lat, lon, dlat, dlon, wrow (float32 row
weights, contrack.py:703-704), thr (float64 per step, already rounded the way the reference's
compare rounds it), gorl, overlap, persistence, twosided, flag (the reference's output, int32).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import refimport  # noqa: E402
from contrack_amd import synth  # noqa: E402
from oracle import cpu_oracle  # noqa: E402


def quantise(a, q=8.0):
    return (np.round(a * q) / q).astype(np.float32)


def ref_slab():
    raw = np.fromfile(os.path.join(refimport.REF_ROOT, "tests/test_data/anom_test.nc"), dtype="<f4",
                      count=11 * 181 * 360, offset=14248).reshape(11, 181, 360)
    lat = np.linspace(90, -90, 181, dtype=np.float32)
    lon = np.arange(360, dtype=np.float32)
    return raw.copy(), lat, lon


def cesm_grid(ny=48, nx=72):
    # Gaussian-like irregular latitudes (needs set_up(force=True) in the reference)
    k = np.arange(ny)
    lat = (90.0 - 180.0 * (k + 0.5) / ny + 0.3 * np.sin(np.pi * k / (ny - 1))).astype(np.float64)
    lon = (np.arange(nx) * (360.0 / nx)).astype(np.float64)
    return lat, lon


CASES = []


def case(name, **kw):
    CASES.append((name, kw))


# --- the reference's own test slab (tests/test_contrack.py:83-91) ------------------------------
case("refslab_fwd", src="ref", threshold=150, gorl=">=", overlap=0.5, persistence=5, twosided=False)
case("refslab_two", src="ref", threshold=150, gorl=">=", overlap=0.5, persistence=5, twosided=True)
case("refslab_lt", src="ref", threshold=-120, gorl="<", overlap=0.3, persistence=3, twosided=True)
# --- synthetic slabs, 2 deg grid ---------------------------------------------------------------
for s in range(4):
    case("syn2deg_s%d" % s, src="syn", T=72, ny=91, nx=180, seed=s, threshold=160, gorl=">=", overlap=0.5,
         persistence=5, twosided=True)
case("syn2deg_fwd", src="syn", T=72, ny=91, nx=180, seed=11, threshold=160, gorl="ge", overlap=0.5,
     persistence=4, twosided=False)
case("syn2deg_ov85", src="syn", T=72, ny=91, nx=180, seed=12, threshold=150, gorl=">", overlap=0.85,
     persistence=3, twosided=True)
case("syn2deg_le", src="syn", T=60, ny=91, nx=180, seed=13, threshold=-90, gorl="<=", overlap=0.5,
     persistence=5, twosided=True, offset=0.0)
case("syn2deg_lt", src="syn", T=60, ny=91, nx=180, seed=14, threshold=-90.0, gorl="lt", overlap=0.4,
     persistence=2, twosided=False, offset=0.0)
case("syn2deg_p1", src="syn", T=40, ny=91, nx=180, seed=15, threshold=170, gorl=">=", overlap=0.0,
     persistence=1, twosided=True)
case("syn2deg_ov1", src="syn", T=40, ny=91, nx=180, seed=16, threshold=140, gorl=">=", overlap=1.0,
     persistence=2, twosided=True)
# busy field: small sigma -> many components, many seam events and chain events
case("busy_s0", src="syn", T=48, ny=61, nx=120, seed=20, threshold=120, gorl=">=", overlap=0.3, persistence=3,
     twosided=True, sigma_deg=4.0, sigma_t=1.0)
case("busy_s1", src="syn", T=48, ny=61, nx=120, seed=21, threshold=100, gorl=">=", overlap=0.25, persistence=2,
     twosided=False, sigma_deg=3.0, sigma_t=1.5)
case("busy_s2", src="syn", T=96, ny=46, nx=90, seed=22, threshold=90, gorl=">=", overlap=0.2, persistence=2,
     twosided=True, sigma_deg=5.0, sigma_t=3.0)
# nx not a multiple of 64 / tiny grids / word-boundary stress
case("odd_65x130", src="syn", dlon=2.75, T=30, ny=65, nx=130, seed=30, threshold=150, gorl=">=", overlap=0.5, persistence=3,
     twosided=True, sigma_deg=8.0)
case("odd_17x64", src="syn", T=24, ny=17, nx=64, seed=31, threshold=130, gorl=">=", overlap=0.5, persistence=2,
     twosided=True, sigma_deg=20.0)
case("odd_9x65", src="syn", dlon=5.5, T=24, ny=9, nx=65, seed=32, threshold=120, gorl=">=", overlap=0.4, persistence=2,
     twosided=True, sigma_deg=30.0)
# 1 deg grid, the BASELINE config-1 parameters on a short slab
case("syn1deg", src="syn", T=40, ny=181, nx=360, seed=40, threshold=160, gorl=">=", overlap=0.5, persistence=5,
     twosided=True)
# white noise: thousands of tiny components per step, exact-0.5 ties, capacity stress
case("noise", src="noise", T=12, ny=46, nx=90, seed=50, threshold=1.0, gorl=">=", overlap=0.5, persistence=2,
     twosided=True)
case("noise_fwd", src="noise", T=10, ny=46, nx=90, seed=51, threshold=0.8, gorl=">=", overlap=0.5, persistence=1,
     twosided=False)
# NaNs (leading NaN planes as produced by calc_anom's rolling mean) and NaN speckles
case("nan_planes", src="syn", T=40, ny=91, nx=180, seed=60, threshold=150, gorl=">=", overlap=0.5, persistence=3,
     twosided=True, nan="planes")
case("nan_speckle", src="syn", T=40, ny=91, nx=180, seed=61, threshold=150, gorl="<", overlap=0.5, persistence=3,
     twosided=True, nan="speckle")
# per-timestep (numpy float64) threshold vector -> float64 compare
case("thr_vector", src="syn", T=40, ny=91, nx=180, seed=70, threshold="vector", gorl=">=", overlap=0.5,
     persistence=3, twosided=True)
# degenerate lengths and fields
# (T=1 is not a golden: the reference itself raises IndexError in set_up, contrack.py:371)
case("T2", src="syn", T=2, ny=31, nx=60, seed=81, threshold=120, gorl=">=", overlap=0.5, persistence=2, twosided=True,
     sigma_deg=12.0, sigma_t=0.5)
case("T3", src="syn", T=3, ny=31, nx=60, seed=82, threshold=110, gorl=">=", overlap=0.5, persistence=2, twosided=True,
     sigma_deg=12.0, sigma_t=0.5)
case("all_bg", src="syn", T=6, ny=31, nx=60, seed=83, threshold=1e9, gorl=">=", overlap=0.5, persistence=2,
     twosided=True, sigma_deg=12.0)
case("all_fg", src="syn", T=6, ny=31, nx=60, seed=84, threshold=-1e9, gorl=">=", overlap=0.5, persistence=2,
     twosided=True, sigma_deg=12.0)
# chain events with PARTIAL box containment (found by a randomized search against the oracle): these
# exercise the per-pixel fold of the bbox-confined seam merges (contrack.py:753-763)
case("chain_a", src="syn", T=120, ny=31, nx=36, seed=1143, threshold=100.0, gorl=">=", overlap=0.5, persistence=2,
     twosided=True, sigma_deg=18.0, sigma_t=1.0)
case("chain_b", src="syn", T=40, ny=31, nx=60, seed=1215, threshold=120.0, gorl=">=", overlap=0.5, persistence=4,
     twosided=False, sigma_deg=6.0, sigma_t=2.0)
case("chain_c", src="syn", T=120, ny=21, nx=40, seed=1308, threshold=120.0, gorl=">=", overlap=0.1, persistence=2,
     twosided=False, sigma_deg=12.0, sigma_t=1.0)
# irregular (CESM-like) latitudes: the reference needs set_up(force=True); dlat = round(mean, 2)
case("cesm_like", src="syn", T=40, ny=48, nx=72, seed=90, threshold=150, gorl=">=", overlap=0.5, persistence=3,
     twosided=True, grid="cesm", sigma_deg=10.0)
# float64 latitudes holding the poles exactly (CESM, MERRA2, new-CDS ERA5, any np.linspace grid): cos(pi/2) = 6e-17 in
# float64, so the pole rows weigh ~1e-13 next to ~1e4 -- row weights spanning ~78 bits, area sums numpy cannot hold exactly
case("f64pole_syn", src="syn", T=40, ny=91, nx=180, seed=95, threshold=120, gorl=">=", overlap=0.5, persistence=3,
     twosided=True, grid="f64", sigma_deg=14.0)
case("f64pole_blocky", src="blocky", T=24, ny=37, nx=72, seed=96, threshold=0.3, gorl=">=", overlap=1.0, persistence=2,
     twosided=True, grid="f64")
case("f64pole_blocky5", src="blocky", T=24, ny=46, nx=96, seed=97, threshold=0.5, gorl=">", overlap=0.5, persistence=1,
     twosided=False, grid="f64")


def levels_for(thr64, gorl):
    """float32 values (inside, outside) hugging the threshold as tightly as float32 allows, so that
    `value <op> thr` is true for `inside` and false for `outside` in the reference's compare."""
    thr64 = np.asarray(thr64, dtype=np.float64)
    f = thr64.astype(np.float32)
    up = np.where(f.astype(np.float64) >= thr64, f, np.nextafter(f, np.float32(np.inf)))      # smallest f32 >= thr
    dn = np.where(f.astype(np.float64) <= thr64, f, np.nextafter(f, np.float32(-np.inf)))     # largest  f32 <= thr
    up_strict = np.where(up.astype(np.float64) > thr64, up, np.nextafter(up, np.float32(np.inf)))
    dn_strict = np.where(dn.astype(np.float64) < thr64, dn, np.nextafter(dn, np.float32(-np.inf)))
    if gorl in (">=", "ge"):
        return up.astype(np.float32), dn_strict.astype(np.float32)
    if gorl in (">", "gt"):
        return up_strict.astype(np.float32), dn.astype(np.float32)
    if gorl in ("<=", "le"):
        return dn.astype(np.float32), up_strict.astype(np.float32)
    return dn_strict.astype(np.float32), up.astype(np.float32)


def decode_input(z):
    """Rebuild the float32 slab a fixture describes (shared with tests/golden_util.py)."""
    if "anom" in z:
        return np.array(z["anom"], dtype=np.float32)
    if "anom_ref" in z:
        return np.array(np.load(os.path.join(HERE, str(z["anom_ref"])))["anom"], dtype=np.float32)
    shape = tuple(int(v) for v in z["shape"])
    n = int(np.prod(shape))
    m = np.unpackbits(z["mask_bits"])[:n].reshape(shape).astype(bool)
    vin = np.broadcast_to(np.asarray(z["v_in"], dtype=np.float32).reshape(-1, 1, 1), shape)
    vout = np.broadcast_to(np.asarray(z["v_out"], dtype=np.float32).reshape(-1, 1, 1), shape)
    a = np.where(m, vin, vout).astype(np.float32)
    if "nan_bits" in z:
        nm = np.unpackbits(z["nan_bits"])[:n].reshape(shape).astype(bool)
        a[nm] = np.nan
    return a


def build_input(kw):
    """Returns (coding dict to store, lat, lon, force)."""
    src = kw["src"]
    if src == "ref":
        _, lat, lon = ref_slab()
        return dict(anom_ref="refslab_input.npz"), lat, lon, False
    T, ny, nx = kw["T"], kw["ny"], kw["nx"]
    if kw.get("grid") == "cesm":
        lat, lon = cesm_grid(ny, nx)
        force = True
    elif kw.get("grid") == "f64":
        lat, lon, force = np.linspace(90.0, -90.0, ny), np.arange(nx) * (360.0 / nx), False
    else:
        lat, lon = synth.grid(ny, nx)
        if "dlon" in kw:
            lon = (np.arange(nx) * kw["dlon"]).astype(np.float32)
        force = False
    if src == "noise":
        a = np.random.default_rng(kw["seed"]).standard_normal((T, ny, nx)).astype(np.float32)
        return dict(anom=quantise(a, 64.0)), lat, lon, force
    if src == "blocky":
        # coarse random field repeated over 3x3 pixels and two steps: plateaus that reach the pole rows, exact overlap ties
        c = np.random.default_rng(kw["seed"]).standard_normal(((T + 1) // 2, (ny + 2) // 3, (nx + 2) // 3)).astype(np.float32)
        a = np.repeat(np.repeat(np.repeat(c, 2, axis=0), 3, axis=1), 3, axis=2)[:T, :ny, :nx].copy()
        return dict(anom=quantise(a, 64.0)), lat, lon, force
    a = synth.smooth_field(T, ny, nx, seed=kw["seed"], offset=kw.get("offset", 35.0),
                           sigma_t=kw.get("sigma_t", 2.0), sigma_deg=kw.get("sigma_deg", 6.0))
    thr = kw["threshold"]
    if isinstance(thr, str) and thr == "vector":
        thr = vector_threshold(T)
    thr64 = np.broadcast_to(np.asarray(thr, dtype=np.float64), (T,))
    op = {">=": np.greater_equal, "ge": np.greater_equal, ">": np.greater, "gt": np.greater,
          "<=": np.less_equal, "le": np.less_equal, "<": np.less, "lt": np.less}[kw["gorl"]]
    m = op(a.astype(np.float64), thr64[:, None, None])
    vin, vout = levels_for(thr64, kw["gorl"])
    coding = dict(shape=np.array(a.shape, dtype=np.int64), mask_bits=np.packbits(m.reshape(-1)),
                  v_in=vin, v_out=vout)
    if kw.get("nan") == "planes":
        nm = np.zeros(a.shape, dtype=bool)
        nm[:2] = True
        nm[-1, :10] = True
        coding["nan_bits"] = np.packbits(nm.reshape(-1))
    elif kw.get("nan") == "speckle":
        nm = np.random.default_rng(1234).random(a.shape) < 0.02
        coding["nan_bits"] = np.packbits(nm.reshape(-1))
    return coding, lat, lon, force


def vector_threshold(T):
    return 150.0 + 20.0 * np.sin(np.arange(T) / 5.0) + 1e-7      # float64, not float32-representable


def main():
    total = 0
    only = set(sys.argv[1:])                 # optional: names of the cases to (re)write
    raw, _, _ = ref_slab()
    if not only:
        np.savez_compressed(os.path.join(HERE, "refslab_input.npz"), anom=raw)
    total += os.path.getsize(os.path.join(HERE, "refslab_input.npz"))
    for name, kw in CASES:
        if only and name not in only:
            continue
        coding, lat, lon, force = build_input(kw)
        a = decode_input(coding)
        T = a.shape[0]
        thr_in = kw["threshold"]
        time = None
        if isinstance(thr_in, str) and thr_in == "vector":
            thr_in = vector_threshold(T)
            # Through the reference's OWN DataArray-threshold branch (contrack.py:648-661): a daily time axis, the threshold as a
            # DataArray over 'dayofyear', `self.ds[variable].groupby('time.dayofyear') >= threshold` (tests/minixr.py provides
            # that groupby comparison and reset_coords, nothing more).  Day of year k+1 gets the k-th value of the vector, i.e.
            # time step k is compared with thr_in[k] -- in float64 (the threshold is a float64 array, contrack.py:650).
            time = (np.datetime64("2001-01-01") + np.arange(T).astype("timedelta64[D]")).astype("datetime64[ns]")
            import minixr
            thr_ref = minixr.DataArray(thr_in, ("dayofyear",), coords={"dayofyear": minixr.DataArray(np.arange(1, T + 1), ("dayofyear",))})
            minixr.install_as_xarray()
            import xarray
            assert isinstance(thr_ref, xarray.DataArray)                 # the reference's isinstance test takes this branch
        else:
            thr_ref = thr_in
        flag, c = refimport.run_reference(a, lat, lon, thr_ref, kw["gorl"], kw["overlap"], kw["persistence"],
                                          kw["twosided"], force=force, time=time)
        dlat, dlon = np.asarray(c._dlat), np.asarray(c._dlon)
        wrow = cpu_oracle.row_weights(lat, c._dlat, c._dlon)
        thr = cpu_oracle.prepare_thresholds(thr_in, T, a.dtype)
        out = dict(lat=lat, lon=lon, dlat=dlat, dlon=dlon, wrow=wrow, thr=thr, gorl=kw["gorl"],
                   overlap=np.float64(kw["overlap"]), persistence=np.int64(kw["persistence"]),
                   twosided=np.bool_(kw["twosided"]), flag_bits=np.packbits(flag.reshape(-1) > 0))
        # the flag array is stored run-length coded per value: ids + packed mask is not enough, so
        # store it as int32 but through its (small) palette: index array uint8/uint16 + palette
        pal, inv = np.unique(flag, return_inverse=True)
        out.pop("flag_bits")
        out["flag_palette"] = pal.astype(np.int32)
        out["flag_index"] = inv.reshape(flag.shape).astype(np.uint8 if len(pal) < 256 else np.uint16)
        out.update(coding)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        sz = os.path.getsize(path)
        total += sz
        f, n = cpu_oracle.run_contrack(a, thr, kw["gorl"], wrow, kw["overlap"], kw["persistence"], kw["twosided"])
        ids = np.unique(flag)
        print("%-14s T=%-3d %3dx%-3d cover=%.3f tracked=%-4d maxid=%-6d oracle_ok=%s  %.1f KB" % (
            name, T, a.shape[1], a.shape[2], float((cpu_oracle.threshold_mask(a, thr, kw["gorl"]) > 0).mean()),
            len(ids) - 1, int(ids.max()), bool(np.array_equal(f, flag)), sz / 1024))
    print("total %.2f MB" % (total / 1e6))


if __name__ == "__main__":
    main()
