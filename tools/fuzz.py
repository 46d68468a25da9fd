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
EDGE = len(sys.argv) > 3 and sys.argv[3] == "edge"

bad, fix, amb = [], 0, 0
with _native.Tracker(0) as t:
    for i in range(first, first + count):
        a, thr, gorl, ov, pers, two = m._edge_case(i) if EDGE else m._random_case(i)
        T, ny, nx = a.shape
        lat = np.linspace(90, -90, ny).astype(np.float32)
        w = cpu_oracle.row_weights(lat, np.float32(180.0 / max(ny - 1, 1)), np.float32(360.0 / nx))
        thrv = cpu_oracle.prepare_thresholds(thr, T)
        want, nw = cpu_oracle.run_contrack(a, thrv, gorl, w, ov, pers, two)
        got, ng = t.track(a, thrv, _native.CMP_OPS[gorl], w, ov, pers, two)
        st = t.stats()
        fix += st["exact_fixups"] > 0; amb += st["ambiguous_decisions"] > 0
        if not (np.array_equal(got, want) and ng == nw):
            bad.append((i, a.shape, gorl, ov, pers, two, st["exact_fixups"], st["ambiguous_decisions"]))
print("cases %d..%d: mismatches %d %s; calls with exact fix-ups %d, with unresolved ambiguity %d" % (first, first + count - 1, len(bad), bad[:5], fix, amb))
