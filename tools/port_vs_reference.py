#!/usr/bin/env python3
"""Build container only (needs /root/reference): wall time of the UNMODIFIED reference run_contrack (imported through
tests/refimport.py + tests/minixr.py) and of oracle/scipy_port.py -- the module bench.py times as `cpu_baseline` on the GPU box,
where the reference's Python cannot travel -- on the same slab, one core, alternating, REPEATS times each.

    python tools/port_vs_reference.py [T]        ->  profiles/port_vs_reference.json

bench.py quotes the file (cpu_baseline.port_vs_reference): the ratio says how far the port's time is from the reference's on
this host; nothing in it is typed in by hand.
"""
import json
import os
import platform
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from contrack_amd import synth  # noqa: E402
from oracle import scipy_port  # noqa: E402
import refimport  # noqa: E402

REPEATS = 7


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 720
    ny, nx = 181, 360
    a = synth.smooth_field(T, ny, nx, seed=0)
    lat, lon = synth.grid(ny, nx)
    w = np.array((111 * np.float32(1.0) * 111 * np.float32(1.0) * np.cos(lat * np.pi / 180))).astype(np.float32)
    thr = np.float32(160.0)
    if not refimport.available():
        raise SystemExit("needs /root/reference")
    refimport.load()
    t_ref, t_port, same = [], [], True
    def one_ref():
        t0 = time.perf_counter()
        f, _ = refimport.run_reference(a, lat, lon, 160.0, ">=", 0.5, 5, True)
        t_ref.append(time.perf_counter() - t0)
        return np.asarray(f)

    def one_port():
        t0 = time.perf_counter()
        f, _ = scipy_port.run_contrack(a, thr, ">=", w, 0.5, 5, True)
        t_port.append(time.perf_counter() - t0)
        return f
    for rep in range(REPEATS):
        if rep % 2 == 0:                                    # (the order alternates too: whoever runs second finds warm caches)
            f_ref = one_ref(); f_port = one_port()
        else:
            f_port = one_port(); f_ref = one_ref()
        same = same and bool(np.array_equal(f_ref, f_port))
        print("repeat %d: reference %.2f s, port %.2f s" % (rep, t_ref[-1], t_port[-1]), flush=True)
    out = dict(slab="%dx%dx%d float32, synth.smooth_field(seed=0), threshold >= 160, overlap 0.5, persistence 5, twosided" % (T, ny, nx),
               repeats=REPEATS, order="alternating; the first of each pair alternates too",
               reference_s=dict(median=statistics.median(t_ref), min=min(t_ref), max=max(t_ref), all=t_ref),
               port_s=dict(median=statistics.median(t_port), min=min(t_port), max=max(t_port), all=t_port),
               port_over_reference=statistics.median(t_port) / statistics.median(t_ref),
               port_over_reference_best=min(t_port) / min(t_ref),
               identical_flags=same, cores=1,
               host="%s, %s, python %s, numpy %s" % (platform.node(), platform.processor() or platform.machine(), platform.python_version(), np.__version__),
               reference="unmodified /root/reference/contrack/contrack.py:583-796 under tests/minixr.py",
               port="oracle/scipy_port.py")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "port_vs_reference.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("port_over_reference", "identical_flags")}))


if __name__ == "__main__":
    main()
