"""Build container only (needs /root/reference): the CPU oracle against the UNMODIFIED reference (run under tests/minixr.py)
on the seeded random cases of tests/test_gpu_parity.py -- beyond the committed goldens.
    python tools/fuzz_oracle_vs_reference.py <first> <count> std|edge
Last run: std 0..1349 (1318 compared) and edge 0..2999 (1074 compared: the reference needs T >= 3 and scalar thresholds
here): no mismatch."""
import sys, time, importlib.util
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refimport
from oracle import cpu_oracle
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
first, count, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
bad = []; t0 = time.time(); n = 0
for i in range(first, first + count):
    a, thr, gorl, ov, pers, two = m._edge_case(i) if mode == "edge" else m._random_case(i)
    T, ny, nx = a.shape
    if T < 3 or ny < 3 or np.ndim(thr) > 0:      # the reference's set_up needs >= 2 steps per axis; scalar thresholds only here
        continue
    lat = np.linspace(90, -90, ny).astype(np.float32); lon = (np.arange(nx) * (360.0 / nx)).astype(np.float32)
    try:
        ref, c = refimport.run_reference(a, lat, lon, thr, gorl, ov, pers, two, force=True)
    except Exception as e:
        bad.append((i, "REF EXC", str(e)[:60])); continue
    w = cpu_oracle.row_weights(lat, c._dlat, c._dlon)
    want, nw = cpu_oracle.run_contrack(a, cpu_oracle.prepare_thresholds(thr, T, a.dtype), gorl, w, ov, pers, two)
    n += 1
    if not np.array_equal(np.asarray(ref).astype(np.int64), want.astype(np.int64)):
        bad.append((i, a.shape, gorl, ov, pers, two))
print("oracle vs reference, %s %d..%d: compared %d, mismatches %d %s  (%.0fs)" % (mode, first, first + count - 1, n, len(bad), bad[:5], time.time() - t0))
