"""Randomized GPU-vs-oracle runs beyond the 36 cases of tests/test_gpu_parity.py::test_randomized_against_oracle
(same generator, other seeds).  python tools/fuzz.py [first] [count]   -- test infrastructure, uses oracle/."""
import os, sys, importlib.util
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from contrack_amd import _native
from oracle import cpu_oracle
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 200)
EDGE = len(sys.argv) > 3 and sys.argv[3] in ("edge", "f64", "thr")
THR = len(sys.argv) > 3 and sys.argv[3] == "thr"
F64 = len(sys.argv) > 3 and sys.argv[3] == "f64"
OPS = {">=": np.greater_equal, ">": np.greater, "<=": np.less_equal, "<": np.less}

bad, fix, amb = [], 0, 0
with _native.Tracker(0) as t:
    for i in range(first, first + count):
        a, thr, gorl, ov, pers, two = m._edge_case(i) if EDGE else m._random_case(i)
        T, ny, nx = a.shape
        lat = np.linspace(90, -90, ny).astype(np.float32)
        w = cpu_oracle.row_weights(lat, np.float32(180.0 / max(ny - 1, 1)), np.float32(360.0 / nx))
        thrv = cpu_oracle.prepare_thresholds(thr, T)
        if F64:
            # float64 data compared in float64 on the device; the oracle gets a float32 surrogate with the same mask
            rng = np.random.default_rng(i)
            a64 = a.astype(np.float64) + rng.choice([0.0, 1e-9, -1e-9, 2.0 ** -30], size=a.shape)
            thr64 = np.broadcast_to(np.asarray(thr, dtype=np.float64), (T,)).copy() + rng.choice([0.0, 1e-9, -1e-9])
            with np.errstate(invalid="ignore"):
                mask = OPS[gorl](a64, thr64[:, None, None])
            want, nw = cpu_oracle.run_contrack(mask.astype(np.float32), np.full(T, 0.5), ">=", w, ov, pers, two)
            got, ng = t.track(a64, thr64, _native.CMP_OPS[gorl], w, ov, pers, two, f64=True)
        elif THR:
            # float32 data within a few ulp of float64 thresholds that float32 cannot represent: the device compares in
            # float32 against an adjusted threshold (adjust_threshold) -- must equal the float64 compare for every value
            rng = np.random.default_rng(i)
            thr64 = rng.choice([0.1, 0.30000000000000004, -2.0 / 3.0, 1e-3, 160.00000001, 1.0 + 2.0 ** -30, -1e-40, 0.0], size=T)
            base = thr64.astype(np.float32)[:, None, None] * np.ones(a.shape, np.float32)
            k = rng.integers(-3, 4, size=a.shape)
            a32 = base.copy()
            for step in range(3):
                a32 = np.where(k > step, np.nextafter(a32, np.float32(np.inf)), a32)
                a32 = np.where(k < -step, np.nextafter(a32, np.float32(-np.inf)), a32)
            a32 = a32.astype(np.float32)
            mask = OPS[gorl](a32.astype(np.float64), thr64[:, None, None])
            want, nw = cpu_oracle.run_contrack(mask.astype(np.float32), np.full(T, 0.5), ">=", w, ov, pers, two)
            got, ng = t.track(a32, thr64, _native.CMP_OPS[gorl], w, ov, pers, two)
        else:
            want, nw = cpu_oracle.run_contrack(a, thrv, gorl, w, ov, pers, two)
            got, ng = t.track(a, thrv, _native.CMP_OPS[gorl], w, ov, pers, two)
        st = t.stats()
        fix += st["exact_fixups"] > 0; amb += st["ambiguous_decisions"] > 0
        if not (np.array_equal(got, want) and ng == nw):
            bad.append((i, a.shape, gorl, ov, pers, two, st["exact_fixups"], st["ambiguous_decisions"]))
print("cases %d..%d: mismatches %d %s; calls with exact fix-ups %d, with unresolved ambiguity %d" % (first, first + count - 1, len(bad), bad[:5], fix, amb))
