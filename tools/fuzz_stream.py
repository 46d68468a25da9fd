"""Seeded random cases (the generator of tests/test_gpu_parity.py) through the streaming entries with random chunk sizes, arrays and
callbacks, float32 and float64, against the one-call path.  python tools/fuzz_stream.py [first] [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from contrack_amd import _native  # noqa: E402
from contrack_amd.contrack import row_weights  # noqa: E402
import test_gpu_parity as tp  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = []
with _native.Tracker(0) as trk:
    for i in range(first, first + count):
        a, thr, gorl, ov, pers, two = tp._random_case(i)
        rng = np.random.default_rng(77 + i)
        T, ny, nx = a.shape
        lat = np.linspace(90, -90, ny).astype(np.float32)
        w = row_weights(lat, np.float32(180.0 / (ny - 1)), np.float32(360.0 / nx))
        f64 = bool(rng.integers(0, 2))
        src = a.astype(np.float64) if f64 else a
        thrv = np.full(T, np.float64(thr) if f64 else np.float64(np.float32(thr)))
        op = _native.CMP_OPS[gorl]
        want, nw = trk.track(src, thrv, op, w, ov, pers, two, f64=f64)
        chunk = int(rng.integers(1, T + 3))
        if rng.integers(0, 2):
            got, ng = trk.track_stream(src, thrv, op, w, ov, pers, two, chunk_steps=chunk)
        else:
            out = np.full((T, ny, nx), -3, dtype=np.int32)

            def rd(t0, nt, dst):
                dst[...] = src[t0:t0 + nt]

            def wr(t0, nt, fl):
                out[t0:t0 + nt] = fl
            _, ng = trk.track_stream(rd, thrv, op, w, ov, pers, two, sink=wr, shape=(T, ny, nx), dtype=src.dtype, chunk_steps=chunk)
            got = out
        if not (np.array_equal(got, want) and ng == nw):
            bad.append((i, T, ny, nx, chunk, f64))
print("stream fuzz %d..%d: problems %d %s" % (first, first + count - 1, len(bad), bad[:10]))
