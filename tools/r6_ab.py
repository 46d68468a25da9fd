#!/usr/bin/env python3
"""Round-6 A/B of environment switches of the one-call pass on a bench workload (GPU box).

    python tools/r6_ab.py [--workload era5_1deg_djf30] [--steps 30] VARIANT [VARIANT ...]

A VARIANT is a comma-separated list of KEY=VALUE environment settings ("base" = none).  The switches are read once per process,
so every variant runs in its own child process on the same slab (the CPU-generated one is cached in /tmp; device-generated
workloads are filled on the device).  Variants are run round-robin `--rounds` times (the board drifts between processes).
Per child: ms per pass over `steps` passes (timing level 0), per-kernel-group event times from 5 extra passes (level 2), the
zero-fill time, n_tracked, and a position-weighted checksum of the flag slab -- which must agree between the variants.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import bench
    from contrack_amd import _native
    wl = bench.WORKLOADS[args.workload]
    T, ny, nx = wl["T"], wl["ny"], wl["nx"]
    trk = _native.Tracker(0)
    nbytes = T * ny * nx * 4
    d_in, d_out = trk.malloc(nbytes), trk.malloc(nbytes)
    if wl.get("device_fill"):
        w = bench.workload_weights(wl)
        bench.device_fill(trk, d_in, wl)
    else:
        cache = "/tmp/r6_slab_%s.npy" % args.workload
        a = np.load(cache)
        _, w = None, bench.workload_weights(wl)
        trk.h2d(d_in, a)
        del a
    thr = np.full(T, np.float64(np.float32(wl["threshold"])))
    op = _native.CMP_OPS[wl["gorl"]]

    def step():
        return trk.track_dev(d_in, T, ny, nx, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"], d_out)
    trk.set_timing(0)
    for _ in range(5):
        n = step()
    trk.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n = step()
    trk.sync()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    cs = trk.checksum_i32(d_out, T * ny * nx)
    trk.set_timing(2)
    acc = {}
    pass
    for _ in range(5):
        step()
        for k, v in trk.timings().items():
            acc[k] = acc.get(k, 0.0) + v / 5
    zf = (0.0, 0)
    st = trk.stats()
    print(json.dumps(dict(ms=ms, n_tracked=n, checksum=[int(x) for x in cs], kernels={k: round(v * 1e3, 1) for k, v in acc.items() if v > 0 and k not in ("total", "h2d", "d2h")},
                          zero_fill_us=round(zf[0] * 1e3, 1), fused=st["fused_pass"], relabel_kernel=st["relabel_kernel"], early_zero=st.get("early_zero_fill"),
                          mask_tries=st["mask_allocations_tried"], mask_ratio=st["mask_ratio_x1000"], mask_check_us=st.get("mask_check_us"))))
    trk.free(d_in)
    trk.free(d_out)
    trk.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="era5_1deg_djf30")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("variants", nargs="*")
    args = ap.parse_args()
    if args.child:
        return child(args)
    import bench
    wl = bench.WORKLOADS[args.workload]
    if not wl.get("device_fill"):
        cache = "/tmp/r6_slab_%s.npy" % args.workload
        if not os.path.exists(cache):
            a, _ = bench.make_slab(wl)
            np.save(cache, a)
    variants = args.variants or ["base"]
    res = {v: [] for v in variants}
    for r in range(args.rounds):
        for v in variants:
            env = dict(os.environ)
            if v != "base":
                for kv in v.split(","):
                    k, val = kv.split("=", 1)
                    env[k] = val
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--workload", args.workload, "--steps", str(args.steps)],
                               env=env, capture_output=True, text=True)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not line:
                print("VARIANT %s FAILED rc=%d\n%s\n%s" % (v, p.returncode, p.stdout[-2000:], p.stderr[-3000:]), flush=True)
                continue
            d = json.loads(line[-1])
            res[v].append(d)
            for ln in sorted(set(x for x in p.stderr.splitlines() if x.startswith("SDDBG")))[:3]:
                print("    " + ln, flush=True)
            print("%-60s %.4f ms  n=%s cs=%s zero=%s us fused=%s rk=%s mask=%s/%s  %s" % (v, d["ms"], d["n_tracked"], d["checksum"][0] % 1000003, d["zero_fill_us"], d["fused"], d["relabel_kernel"],
                                                                                      d["mask_tries"], d["mask_ratio"], d["kernels"]), flush=True)
    print("---- summary (min / median ms per pass)")
    sums = set()
    for v in variants:
        xs = sorted(d["ms"] for d in res[v])
        if xs:
            print("%-60s min %.4f  med %.4f" % (v, xs[0], xs[len(xs) // 2]))
        for d in res[v]:
            sums.add((d["n_tracked"], tuple(d["checksum"])))
    print("distinct (n_tracked, checksum) over all runs: %d %s" % (len(sums), "OK" if len(sums) == 1 else "MISMATCH " + str(sums)))


if __name__ == "__main__":
    main()
