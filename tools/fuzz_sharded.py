"""Time shards driven in ONE process: N handles on one GPU play N ranks.  Random cases, random shard boundaries (shards of
one step included): the product path ctk_track_sharded_* (threads + in-process communicator) and the staged API with the
host resolver, compared with the single-call result.  python tools/fuzz_sharded.py [first] [count]"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from contrack_amd import _native  # noqa: E402
from oracle import cpu_oracle  # noqa: E402

spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 100)


from shard_inproc import sharded, sharded_threads  # noqa: E402

bad = []
trks = [_native.Tracker(0) for _ in range(6)]
ref = _native.Tracker(0)
for i in range(first, first + count):
    a, thr, gorl, ov, pers, two = m._random_case(i) if i % 3 else m._edge_case(i)
    T, ny, nx = a.shape
    if T < 2:
        continue
    rng = np.random.default_rng(7000 + i)
    n = int(rng.integers(2, min(6, T) + 1))
    inner = np.sort(rng.choice(np.arange(1, T), size=n - 1, replace=False))
    cuts = [0] + [int(v) for v in inner] + [T]
    lat = np.linspace(90, -90, ny).astype(np.float32)
    w = cpu_oracle.row_weights(lat, np.float32(180.0 / max(ny - 1, 1)), np.float32(360.0 / nx))
    thrv = cpu_oracle.prepare_thresholds(thr, T)
    op = _native.CMP_OPS[gorl]
    want, nw = ref.track(a, thrv, op, w, ov, pers, two)
    fixups = ref.stats()["exact_fixups"]
    try:
        got, ng, _ = sharded_threads(trks[:n], a, thrv, op, w, ov, pers, two, cuts)
        if not (np.array_equal(got, want) and ng == nw):
            bad.append((i, a.shape, cuts, "product", ov))
        got, ng, info = sharded(trks[:n], a, thrv, op, w, ov, pers, two, cuts)
        if not (np.array_equal(got, want) and ng == nw) and not fixups:
            bad.append((i, a.shape, cuts, "staged", ov))
    except Exception as e:                                    # noqa: BLE001
        bad.append((i, cuts, "EXC " + str(e)[:80]))
print("sharded fuzz %d..%d: problems %d %s" % (first, first + count - 1, len(bad), bad[:6]))
