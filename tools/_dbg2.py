import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from contrack_amd import _native, synth
from oracle import cpu_oracle
import importlib.util
spec = importlib.util.spec_from_file_location("tgp", "/root/repo/tests/test_gpu_parity.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
i = int(sys.argv[1])
a, thr, gorl, ov, pers, two = m._random_case(i)
T, ny, nx = a.shape
lat = np.linspace(90, -90, ny).astype(np.float32)
w = cpu_oracle.row_weights(lat, np.float32(180.0/(ny-1)), np.float32(360.0/nx))
thrv = cpu_oracle.prepare_thresholds(thr, T)
want, nw = cpu_oracle.run_contrack(a, thrv, gorl, w, ov, pers, two)
with _native.Tracker(0) as t:
    got, ng = t.track(a, thrv, _native.CMP_OPS[gorl], w, ov, pers, two)
    t.set_device_resolve(False)
    goth, ngh = t.track(a, thrv, _native.CMP_OPS[gorl], w, ov, pers, two)
print("oracle n", nw, "dev n", ng, "host n", ngh, "dev==oracle", np.array_equal(got, want), "host==oracle", np.array_equal(goth, want), "dev==host", np.array_equal(got, goth))
d = np.argwhere(got != want)
print("diff pixels", len(d), d[:5], "weights first/last", w[:2], w[-2:])
if len(d):
    tt = d[0][0]
    print("t", tt, "oracle ids there", np.unique(want[tt]), "dev ids", np.unique(got[tt]))
# blob info via staged API
from tests import cpu_tables
with _native.Tracker(0) as t:
    d_in = t.malloc(a.nbytes); t.h2d(d_in, a)
    t.shard_label2d(d_in, T, ny, nx, thrv, _native.CMP_OPS[gorl], w, False)
    t.shard_overlap()
    blob = t.shard_tables()
    res = _native.resolve([blob], ov, two)
    print("host resolver info:", res.info())
    t.free(d_in)
