#!/usr/bin/env python3
"""bench.py -- timesteps/s labelled+tracked on synthetic ERA5-like Z500 anomaly slabs (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload era5_1deg_djf30|era5_025deg|...]

A "step" is one pass of the hot path (threshold -> 2-D labelling with longitude wrap -> overlap filter ->
3-D tracking -> persistence -> flag) over one batch: the whole (T, ny, nx) slab of the workload, already
resident in HBM.  For N > 1 (launched as the contract says, one rank per GPU; only the launcher comes from torch) the time
axis is sharded across ranks -- weak scaling by default: every GPU holds one member of the workload (2707 steps at 1 deg),
the members are concatenated on the time axis (BASELINE.json configs[4] layout) and tracked as ONE slab of N x T steps by
ctk_track_sharded_* (one-timestep halo + boundary records over RCCL, shard-local resolver); `--scaling strong` splits the
single-GPU slab instead -- see bench_dist.py (the bench leg) and contrack_amd/dist.py (the product driver).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from contrack_amd import _native, synth  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]: ERA5 Z500 anomaly 1 deg, 2707 DJF daily steps, thr 160 gpm, overlap 0.5, persistence 5
    "era5_1deg_djf30": dict(T=2707, ny=181, nx=360, threshold=160.0, gorl=">=", overlap=0.5, persistence=5, twosided=True),
    # BASELINE.json configs[0]: the reference's own CPU-runnable case
    "era5_1deg_90": dict(T=90, ny=181, nx=360, threshold=160.0, gorl=">=", overlap=0.5, persistence=5, twosided=True),
    # 0.25 deg, 6-hourly (a 480-step window of configs[2]; persistence 20 steps = 5 days)
    "era5_025deg_480": dict(T=480, ny=721, nx=1440, threshold=160.0, gorl=">=", overlap=0.5, persistence=20, twosided=True),
}
WORKLOADS["era5_025deg_10yr"] = dict(T=14600, ny=721, nx=1440, threshold=160.0, gorl=">=", overlap=0.5, persistence=20, twosided=True,
                                     device_fill=True)      # BASELINE.json configs[2]: 60.6 GB in + 60.6 GB out, generated on the device
WORKLOADS["era5_025deg_1k"] = dict(T=1000, ny=721, nx=1440, threshold=160.0, gorl=">=", overlap=0.5, persistence=20, twosided=True, device_fill=True)
WORKLOADS["era5_025deg_2k"] = dict(T=2000, ny=721, nx=1440, threshold=160.0, gorl=">=", overlap=0.5, persistence=20, twosided=True, device_fill=True)
# BASELINE.json configs[4]: CESM-LE Z500, 40 members x 10 950 daily steps concatenated on the time axis, 192 x 288, float64 irregular latitudes
# (the reference needs set_up(force=True): dlat = round(mean spacing, 2), contrack.py:357-370); 96.9 GB in + 96.9 GB out on one GPU,
# every member its own device-generated field
WORKLOADS["cesm_le_40x30yr"] = dict(T=438000, ny=192, nx=288, threshold=160.0, gorl=">=", overlap=0.5, persistence=5, twosided=True,
                                    device_fill=True, members=40, cesm_grid=True)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def row_weights(lat, dlat, dlon):
    # contrack/contrack.py:703-704
    weight_lat = np.cos(lat * np.pi / 180)
    return np.array((111 * dlat * 111 * dlon * weight_lat)).astype(np.float32)


def make_slab(wl, seed=0):
    a = synth.smooth_field(wl["T"], wl["ny"], wl["nx"], seed=seed)
    lat, lon = synth.grid(wl["ny"], wl["nx"])
    dlat = np.float32(180.0 / (wl["ny"] - 1))
    dlon = np.float32(360.0 / wl["nx"])
    return a, row_weights(lat, dlat, dlon)


def cesm_latitudes(ny):
    """Gaussian-like float64 latitudes with irregular spacing (tests/golden/make_golden.py::cesm_grid, tests/test_gpu_fullsize_cesm.py)"""
    k = np.arange(ny)
    return (90.0 - 180.0 * (k + 0.5) / ny + 0.3 * np.sin(np.pi * k / (ny - 1))).astype(np.float64)


def workload_weights(wl):
    """row weights as the reference's set_up + contrack.py:703-704 give them for the workload's grid"""
    ny, nx = wl["ny"], wl["nx"]
    if wl.get("cesm_grid"):
        lat = cesm_latitudes(ny)
        dlat = np.float64(round(float(np.abs(np.diff(lat)).mean()), 2))       # contrack.py:365 (force=True)
        return row_weights(lat, dlat, np.float64(360.0 / nx))
    lat, _ = synth.grid(ny, nx)
    return row_weights(lat, np.float32(180.0 / (ny - 1)), np.float32(360.0 / nx))


def device_fill(trk, d_in, wl):
    """the workload's slab generated on the device; `members` > 1: every member of the concatenated slab its own field"""
    import ctypes as C
    T, ny, nx = wl["T"], wl["ny"], wl["nx"]
    m = int(wl.get("members", 1))
    if m <= 1:
        trk.synth_fill(d_in, T, ny, nx, seed=0)
        return
    per = (T + m - 1) // m
    for q, tb in enumerate(range(0, T, per)):
        trk.synth_fill(C.c_void_p(d_in.value + tb * ny * nx * 4), min(per, T - tb), ny, nx, seed=100 + q)


def pmc_traffic_source(kernel, workload):
    """(bytes, where they come from) -- see pmc_traffic"""
    import glob
    best = (None, None)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            doc = json.load(open(path))
            k = doc["kernels"]
        except Exception:
            continue
        if ("workload " + workload) not in doc.get("note", ""):
            continue
        for name, v in k.items():
            if name.split("<")[0] == kernel:
                best = (v["fetch_bytes_corrected"] + v["write_bytes"],
                        "profiles/%s: committed rocprofv3 --pmc capture of this workload (FETCH_SIZE x 2 + WRITE_SIZE per launch, tools/profile.sh); "
                        "NOT measured in this run" % os.path.basename(path))
    return best


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC capture of this very workload
    (profiles/*_pmc.json, tools/profile.sh + tools/summarize_profile.py; FETCH_SIZE x2 on gfx950 as
    MI355X_MICROARCH.md prescribes, plus WRITE_SIZE).  None if no capture is committed for the workload."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            doc = json.load(open(path))
            k = doc["kernels"]
        except Exception:
            continue
        if ("workload " + workload) not in doc.get("note", ""):
            continue
        for name, v in k.items():
            if name.split("<")[0] == kernel:
                best = v["fetch_bytes_corrected"] + v["write_bytes"]
    return best


def pmc_child(args):
    """hidden mode (--pmc-child): a few passes on the slab the parent cached, for a rocprofv3 --pmc pass around this process"""
    wl = WORKLOADS[args.workload]
    T, ny, nx = wl["T"], wl["ny"], wl["nx"]
    trk = _native.Tracker(int(os.environ.get("LOCAL_RANK", "0")))
    d_in, d_out = trk.malloc(T * ny * nx * 4), trk.malloc(T * ny * nx * 4)
    if wl.get("device_fill"):
        w = workload_weights(wl)
        device_fill(trk, d_in, wl)
    else:
        a = np.load(args.pmc_child)
        w = workload_weights(wl)
        trk.h2d(d_in, a)
        del a
    thr = np.full(T, np.float64(np.float32(wl["threshold"])))
    trk.set_timing(0)
    for _ in range(6):                                      # 2 set-up calls + 4 passes: launches below half of the largest are left out of the mean
        trk.track_dev(d_in, T, ny, nx, thr, _native.CMP_OPS[wl["gorl"]], w, wl["overlap"], wl["persistence"], wl["twosided"], d_out)
    trk.sync()
    trk.free(d_in)
    trk.free(d_out)
    trk.close()


def live_pmc_traffic(workload, a, kernels, timeout_s=150):
    """HBM bytes per launch of `kernels`, MEASURED IN THIS RUN: two child processes of this bench.py on the same slab under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `... WRITE_SIZE` (separate passes: the TCC has four counter slots), corrected as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts wide coalesced reads at half their bytes: x 2;
    KB -> bytes), per launch, launches below half of the largest left out (placement check / tuning launches on windows).
    Returns ({kernel: bytes}, note) or (None, why-not)."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    tmp = tempfile.mkdtemp(prefix="ctk_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        slab = os.path.join(tmp, "slab.npy")
        if a is not None:
            np.save(slab, a)
        vals = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "r", "--",
                   sys.executable, os.path.abspath(__file__), "--workload", workload, "--pmc-child", slab]
            try:
                p = subprocess.run(cmd, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s, env=dict(os.environ, TMPDIR=tmp))
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s did not finish within %d s" % (counter, timeout_s)
            if p.returncode != 0:
                return None, "rocprofv3 --pmc %s exited with %d" % (counter, p.returncode)
            acc = collections.defaultdict(list)
            for path in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(path)):
                    if r["Counter_Name"] == counter:
                        acc[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]))
            vals[counter] = acc
        res = {}
        for k in kernels:
            f, w = vals["FETCH_SIZE"].get(k, []), vals["WRITE_SIZE"].get(k, [])
            if not f or not w:
                continue
            f = [x for x in f if x >= 0.5 * max(f)]
            w = [x for x in w if x >= 0.5 * max(w)]
            res[k] = dict(bytes=2.0 * 1024.0 * sum(f) / len(f) + 1024.0 * sum(w) / len(w), fetch_bytes_corrected=2.0 * 1024.0 * sum(f) / len(f),
                          write_bytes=1024.0 * sum(w) / len(w), launches=min(len(f), len(w)))
        if not res:
            return None, "no counter rows for %s" % ", ".join(kernels)
        return res, ("measured in THIS run: two child processes of bench.py (--pmc-child, the same slab) under rocprofv3 --kernel-trace --pmc FETCH_SIZE / "
                     "WRITE_SIZE (separate passes), FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 per launch (gfx950 correction of MI355X_MICROARCH.md)")
    except Exception as e:                                   # (a measurement aid must never take the bench line down)
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def stream_ceiling(trk, d_in, d_out, nbytes, per):
    try:
        rd = trk.stream_ceiling(d_in, nbytes, write=0)
        wr_order, wr_xcd = trk.stream_ceiling(d_out, nbytes, write=1), trk.stream_ceiling(d_out, nbytes, write=2)
    except _native.ContrackHipError as e:
        return dict(error=str(e))
    wr = min(wr_order, wr_xcd)
    out = dict(plain_load_stream_gbs=nbytes / rd / 1e6, plain_store_stream_gbs=nbytes / wr / 1e6, plain_load_ms=rd, plain_store_ms=wr,
               plain_store_stream_launch_order_gbs=nbytes / wr_order / 1e6, plain_store_stream_eighth_per_xcd_gbs=nbytes / wr_xcd / 1e6,
               note="k_stream_load / k_stream_store (ctk_debug_stream_ceiling): 16-byte non-temporal loads of the slab / stores over the flag slab, "
                    "nothing else; the store stream in two chunk orders (launch order -- rounds 1-6a quoted this one -- and one contiguous eighth "
                    "of the slab per XCD), plain_store_* = the faster")
    if per.get("k_threshold", 0) > 0:
        out["k_threshold_of_plain_load_stream"] = rd / per["k_threshold"]
    if per.get("k_relabel", 0) > 0:
        out["k_relabel_of_plain_store_stream"] = wr / per["k_relabel"]
    return out


def cpu_baseline(wl, a, w, budget_s=20.0):
    """The CPU restatement of the reference path (oracle/scipy_port.py: same scipy.ndimage / numpy call
    sequence as contrack.py:646-796, one core) timed on a bounded sample of the same workload."""
    from oracle import scipy_port
    n = wl["T"] if wl["ny"] * wl["nx"] < 100000 else min(wl["T"], 240)
    thr = np.float32(wl["threshold"])
    t0 = time.perf_counter()
    scipy_port.run_contrack(a[:n], thr, wl["gorl"], w, wl["overlap"], wl["persistence"], wl["twosided"])
    dt = time.perf_counter() - t0
    out = dict(value=n / dt, unit="timesteps/s", cores=1, kind="port",
               sample="first %d of %d steps of the same slab, scipy.ndimage/numpy port of contrack.py:646-796 "
                      "(oracle/scipy_port.py), %.2f s" % (n, wl["T"], dt))
    # how the port's wall time relates to the unmodified reference's: MEASURED in the build container by tools/port_vs_reference.py
    # (alternating repeats on one slab, one core) and committed; quoted here, not typed in
    try:
        pv = json.load(open(os.path.join(ROOT, "profiles", "port_vs_reference.json")))
        out["port_vs_reference"] = dict(port_over_reference_median=pv["port_over_reference"], port_over_reference_best=pv.get("port_over_reference_best"),
                                        reference_s=pv["reference_s"]["median"], port_s=pv["port_s"]["median"], repeats=pv["repeats"], slab=pv["slab"],
                                        identical_flags=pv["identical_flags"], source="profiles/port_vs_reference.json (tools/port_vs_reference.py, build container)")
    except (OSError, KeyError, ValueError):
        out["port_vs_reference"] = None
    return out


def concurrent_members(wl, a, thr, op, w, nh, steps):
    """nh handles, each tracking its own copy of the slab `steps` times, all at once"""
    import threading
    T, ny, nx = wl["T"], wl["ny"], wl["nx"]
    nbytes = T * ny * nx * 4
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    trks = [_native.Tracker(dev) for _ in range(nh)]
    bufs = []
    for t in trks:
        di, do = t.malloc(nbytes), t.malloc(nbytes)
        if a is not None:
            t.h2d(di, a)
        else:
            t.synth_fill(di, T, ny, nx, seed=0)
        t.set_timing(0)
        for _ in range(2):                                    # set-up: work spaces (1st call), placement check / write-kernel tuning (2nd)
            t.track_dev(di, T, ny, nx, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"], do)
        bufs.append((di, do))
    for t in trks:
        t.sync()
    start = threading.Barrier(nh + 1)
    res = [None] * nh

    def run(i):
        t, (di, do) = trks[i], bufs[i]
        start.wait()
        for _ in range(steps):
            res[i] = t.track_dev(di, T, ny, nx, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"], do)
        t.sync()
    th = [threading.Thread(target=run, args=(i,)) for i in range(nh)]
    for x in th:
        x.start()
    start.wait()
    t0 = time.perf_counter()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    for t, (di, do) in zip(trks, bufs):
        t.free(di)
        t.free(do)
        t.close()
    return dict(handles=nh, passes_each=steps, ms_per_slab=dt * 1e3 / (nh * steps), timesteps_per_s=T * nh * steps / dt, n_tracked=res[0],
                note="independent slabs on %d handles / streams / host threads of ONE GPU; not the headline value" % nh)


def secondary_025deg(steps, warmup, budget_steps=240):
    """The grid the north_star's >= 50x is stated on, timed in the same driver-run process: a device-generated 480 x 721 x 1440
    window of BASELINE.json configs[2] (6-hourly, persistence 20 steps), its own roofline (HIP events around the streaming kernels),
    and the CPU leg on the first `budget_steps` steps of the very same slab (downloaded), whose flags must equal the GPU's."""
    from oracle import scipy_port
    name = "era5_025deg_480"
    wl = WORKLOADS[name]
    T, ny, nx = wl["T"], wl["ny"], wl["nx"]
    px = T * ny * nx
    trk = _native.Tracker(int(os.environ.get("LOCAL_RANK", "0")))
    d_in, d_out = trk.malloc(px * 4), trk.malloc(px * 4)
    try:
        lat, _ = synth.grid(ny, nx)
        w = row_weights(lat, np.float32(180.0 / (ny - 1)), np.float32(360.0 / nx))
        trk.synth_fill(d_in, T, ny, nx, seed=0)
        thr = np.full(T, np.float64(np.float32(wl["threshold"])))
        op = _native.CMP_OPS[wl["gorl"]]
        trk.set_timing(1)

        def step(n=T):
            return trk.track_dev(d_in, n, ny, nx, thr[:n], op, w, wl["overlap"], wl["persistence"], wl["twosided"], d_out)
        for _ in range(max(warmup, 2) + 1):
            n_tracked = step()
        trk.sync()
        trk.timing_sums(reset=True)
        k = max(steps, 8)
        t0 = time.perf_counter()
        for _ in range(k):
            n_tracked = step()
        trk.sync()
        dt = time.perf_counter() - t0
        per, _ = trk.timing_sums(reset=True)
        ms = dt * 1e3 / k
        alg = {"k_threshold": 4.0 * px, "k_relabel": 4.0 * px}
        kern = max(alg, key=lambda q: per.get(q, 0.0))
        ach = alg[kern] / (per[kern] * 1e-3) / 1e9 if per.get(kern, 0) > 0 else 0.0
        rk = {5: "k_relabel_v5", 4: "k_relabel_v4"}.get(trk.stats().get("relabel_kernel", 4), "k_relabel")
        kname = {"k_threshold": "k_threshold_v4" if os.environ.get("CTK_THRESHOLD") == "4" else "k_threshold_v7", "k_relabel": rk}[kern]
        out = dict(workload="%s: %dx%dx%d float32 (device-generated), threshold %s %g, overlap %g, persistence %d, twosided %s" % (
                       name, T, ny, nx, wl["gorl"], wl["threshold"], wl["overlap"], wl["persistence"], wl["twosided"]),
                   value=T / (ms * 1e-3), unit="timesteps/s", ms_per_step=ms, steps=k, n_tracked=n_tracked,
                   path_effective_gbs=8.0 * px / (ms * 1e-3) / 1e9,
                   roofline=dict(bound="hbm", kernel=kname, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                                 traffic=pmc_traffic_source(kname, name)[0], traffic_source=pmc_traffic_source(kname, name)[1],
                                 algorithmic_bytes_per_launch=alg[kern], avg_kernel_ms=per.get(kern),
                                 other_streaming_kernel={q: dict(achieved=alg[q] / (per[q] * 1e-3) / 1e9, avg_kernel_ms=per[q])
                                                         for q in alg if q != kern and per.get(q, 0) > 0}))
        out["workload_stats"] = trk.stats()
        # CPU leg: the first n steps of the same slab, downloaded; the GPU's flags of those n steps alongside
        n = min(budget_steps, T)
        a = np.empty((n, ny, nx), dtype=np.float32)
        trk.d2h(a, d_in)
        trk.set_timing(0)
        n_gpu = step(n)
        f_gpu = np.empty((n, ny, nx), dtype=np.int32)
        trk.d2h(f_gpu, d_out)
        t1 = time.perf_counter()
        f_cpu = scipy_port.run_contrack(a, np.float32(wl["threshold"]), wl["gorl"], w, wl["overlap"], wl["persistence"], wl["twosided"])
        dc = time.perf_counter() - t1
        f_cpu = f_cpu[0] if isinstance(f_cpu, tuple) else f_cpu
        out["cpu_baseline"] = dict(value=n / dc, unit="timesteps/s", cores=1, kind="port",
                                   sample="first %d of %d steps of the same device-generated slab, scipy.ndimage/numpy port of contrack.py:646-796 "
                                          "(oracle/scipy_port.py), %.2f s" % (n, T, dc),
                                   flags_equal_gpu=bool(np.array_equal(np.asarray(f_cpu), f_gpu)), n_tracked_gpu_same_sample=n_gpu)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        out["coverage_first_%d_steps" % n] = float((a >= np.float32(wl["threshold"])).mean())
        return out
    finally:
        trk.free(d_in)
        trk.free(d_out)
        trk.close()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, the environment a launcher would
    set), pass rank 0's line through, exit non-zero if any rank does.  Never a 1-GPU line for --gpus N."""
    import socket
    import subprocess
    backend = os.environ.get("CTK_DIST_BACKEND", "rccl")
    ndev = _native.device_count()
    if backend == "rccl" and ndev < n:
        print("bench.py: --gpus %d asked for, %d HIP device(s) visible: RCCL wants one device per rank "
              "(CTK_DIST_BACKEND=shm runs several ranks on one device for plumbing tests)" % (n, ndev), file=sys.stderr)
        return 2
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CTK_LAUNCH_PID=str(os.getpid()), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else sys.stderr))
    rc, t_fail = 0, None
    while any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() not in (None, 0) and rc == 0:
                rc, t_fail = p.returncode, time.time()
        # a failed rank tells the others through the communicator (they give up within its deadline); whoever still runs long
        # after that is stopped by PID
        if t_fail is not None and time.time() - t_fail > float(os.environ.get("CTK_COMM_TIMEOUT_S", "120")) + 30:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    for p in procs:
        if p.returncode != 0 and rc == 0:
            rc = p.returncode
    return rc if rc >= 0 else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="era5_1deg_djf30")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed passes that time every kernel group (profiling runs)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 0.25 deg block (`secondary_025deg`) of the default N = 1 line")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = one member of the workload per GPU, concatenated on the time axis (default); "
                         "strong = the single-GPU slab split over the GPUs")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed capture under profiles/ instead of two rocprofv3 --pmc child runs")
    ap.add_argument("--no-parity-check", action="store_true",
                    help="N > 1: skip the in-run parity check (rank 0 tracks the concatenated slab with one call and compares checksums)")
    ap.add_argument("--strong-steps", type=int, default=-1,
                    help="N > 1: timesteps of the device-generated 0.25 deg slab of the strong-scaling block (0 = skip; default: the 10-year "
                         "slab of BASELINE.json configs[2], 14600 steps -- the size the north_star's >= 6x is stated on --, falling back to "
                         "2000 steps if rank 0's GPU cannot hold it for the one-GPU reference time)")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    if args.pmc_child:
        return pmc_child(args)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))                     # no launcher: the ranks are started here
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to print a line for another job size" % (
            args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if world > 1 or os.environ.get("CTK_FORCE_DIST") == "1":
        import bench_dist
        sys.exit(bench_dist.bench_main(args, wl, WORKLOADS, HBM_PEAK_GBS, cpu_baseline=cpu_baseline, pmc_traffic=pmc_traffic) or 0)

    T, ny, nx = wl["T"], wl["ny"], wl["nx"]
    thr = np.full(T, np.float64(np.float32(wl["threshold"])))
    op = _native.CMP_OPS[wl["gorl"]]
    trk = _native.Tracker(int(os.environ.get("LOCAL_RANK", "0")))
    nbytes = T * ny * nx * 4
    d_in = trk.malloc(nbytes)
    d_out = trk.malloc(nbytes)
    if wl.get("device_fill"):
        # slabs beyond host RAM: deterministic on-device generator (ctk_synth_fill), no CPU baseline / coverage
        a = None
        w = workload_weights(wl)
        device_fill(trk, d_in, wl)
        args.no_cpu_baseline = True
    else:
        a, w = make_slab(wl)
        trk.h2d(d_in, a)
    trk.set_timing(1)          # timed region: HIP events around the two streaming kernels only (the roofline kernels)

    def step():
        return trk.track_dev(d_in, T, ny, nx, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"], d_out)

    # set-up, before the W warm-up steps: the handle's first two calls on a slab size its work space (1st) and check where its bit mask
    # lies / which chunk order the write kernel likes (2nd) -- once per handle, never inside the timed region whatever W is
    for _ in range(2):
        n_tracked = step()
    for _ in range(args.warmup):
        n_tracked = step()
    trk.sync()
    # (level-1 timing: an event pair around ONE of the two streaming kernels in every second pass; the library sums the times,
    # they are read once after the loop -- fetching them after every pass was ~10 us of interpreter time with the GPU idle)
    trk.timing_sums(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_tracked = step()
    trk.sync()
    dt = time.perf_counter() - t0
    ms_per_step = dt * 1e3 / args.steps
    value = T * args.steps / dt
    per, nmeas = trk.timing_sums(reset=True)
    last = trk.timings()                              # (host-side timers of the last pass: informational)
    per.update({k: last[k] for k in ("host_seam_driver", "total", "d2h", "h2d")})
    # per-group kernel times of the small kernels: a few extra, untimed passes with events around every group (each
    # event record is a command of its own and would stretch the timed passes)
    trk.set_timing(2)
    extra, acc2 = (0 if args.no_extra else 5), {}
    for _ in range(extra):
        step()
        for k, v in trk.timings().items():
            acc2[k] = acc2.get(k, 0.0) + v / extra
    for k, v in acc2.items():
        if k not in ("k_threshold", "k_relabel", "total", "d2h", "h2d", "host_seam_driver"):
            per[k] = v
    for k in ("k_threshold", "k_relabel"):      # fewer than four timed passes: a streaming kernel may have had no event pair
        if nmeas.get(k, 0) == 0:
            if not acc2.get(k):
                for _ in range(4):
                    step()
                    v = trk.timings().get(k, 0.0)
                    if v > 0:
                        acc2[k] = v
            per[k] = acc2.get(k, 0.0)
    # roofline of the dominant kernel (by average duration, HIP events on the library's stream)
    px = T * ny * nx
    alg_bytes = {"k_threshold": 4.0 * px, "k_relabel": 4.0 * px}          # float32 read once / int32 written once
    kern = max(alg_bytes, key=lambda k: per.get(k, 0.0))
    achieved = alg_bytes[kern] / (per[kern] * 1e-3) / 1e9 if per.get(kern, 0) > 0 else 0.0
    rk = {5: "k_relabel_v5", 4: "k_relabel_v4"}.get(trk.stats().get("relabel_kernel", 4), "k_relabel")
    kname = {"k_threshold": "k_threshold_v4" if os.environ.get("CTK_THRESHOLD") == "4" else "k_threshold_v7", "k_relabel": rk}[kern]
    coverage = float((a >= np.float32(wl["threshold"])).mean()) if a is not None else None      # (device-generated slabs: filled in at the end)
    traffic, traffic_source = pmc_traffic_source(kname, args.workload)
    live_pmc = None
    if not args.no_live_pmc and not args.no_extra and nbytes <= (8 << 30):
        # HBM traffic of the two streaming kernels measured in this very run (round-5 verdict weak #5); the committed capture stays the fallback
        live, why = live_pmc_traffic(args.workload, a, ["k_threshold_v7", rk])
        if live and kname in live:
            traffic, traffic_source = live[kname]["bytes"], why
            live_pmc = live
        else:
            traffic_source = (traffic_source or "") + " [live rocprofv3 --pmc measurement not available: %s]" % why
    out = dict(metric="timesteps/sec labeled+tracked", value=value, unit="timesteps/s", n_gpus=1, steps=args.steps,
               warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype="f32 compare / int32 labels / int64 exact areas", data="synthetic",
               config=dict(workload="%s: %dx%dx%d float32, threshold %s %g, overlap %g, persistence %d, twosided %s" % (
                   args.workload, T, ny, nx, wl["gorl"], wl["threshold"], wl["overlap"], wl["persistence"], wl["twosided"]),
                   parallelism="1 GPU", n_tracked=n_tracked,
                   coverage=coverage),
               roofline=dict(bound="hbm", kernel=kname, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                             traffic=traffic, traffic_source=traffic_source, algorithmic_bytes_per_launch=alg_bytes[kern],
                             avg_kernel_ms=per.get(kern),
                             other_streaming_kernel={k: dict(achieved=alg_bytes[k] / (per[k] * 1e-3) / 1e9, avg_kernel_ms=per[k])
                                                     for k in alg_bytes if k != kern and per.get(k, 0) > 0}),
               kernels_ms=per, workload_stats=trk.stats(), pmc_live=live_pmc,
               # what a PLAIN 16-byte non-temporal stream reaches on this board, measured here on the same two buffers (best of 5): the write
               # kernel cannot beat the store stream, the threshold kernel not the load stream -- `frac` above stays against the 8 TB/s peak
               stream_ceiling=stream_ceiling(trk, d_in, d_out, nbytes, per),
               path_effective_gbs=8.0 * px / (ms_per_step * 1e-3) / 1e9,
               # the two pixel-streaming kernels together move the path's algorithmic 8 B per pixel
               streaming=dict(bytes=8.0 * px, ms=per.get("k_threshold", 0.0) + per.get("k_relabel", 0.0),
                              achieved_gbs=8.0 * px / ((per.get("k_threshold", 0.0) + per.get("k_relabel", 0.0)) * 1e-3) / 1e9
                              if per.get("k_threshold", 0) + per.get("k_relabel", 0) > 0 else 0.0,
                              share_of_pass=(per.get("k_threshold", 0.0) + per.get("k_relabel", 0.0)) / ms_per_step))
    # what a run_contrack() caller pays: host numpy in -> host numpy out (ctk_track_f32: H2D, kernels, D2H); never `value`
    if a is not None and not args.no_extra:
        trk.set_timing(0)
        trk.track(a, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"])              # warm-up: pinned bounce buffers, device copies

        def e2e_leg(keep_results, reps=4):
            # keep_results: every result stays alive (each call gets a fresh array: first-touch page faults under the D2H copy);
            # else the result is dropped before the next call, as a loop over ensemble members does after writing it out -- its memory
            # is recycled by the binding (registered with HIP, one DMA)
            e2e, keep, lib = 0.0, [], {}
            for _ in range(reps):
                t1 = time.perf_counter()
                f = trk.track(a, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"])[0]
                e2e += (time.perf_counter() - t1) / reps
                for k, v in trk.timings().items():
                    lib[k] = lib.get(k, 0.0) + v / reps
                if keep_results:
                    keep.append(f)
                del f
            return dict(ms_per_call=e2e * 1e3, timesteps_per_s=T / e2e, h2d_ms=lib["h2d"], h2d_gb_per_s=4.0 * px / lib["h2d"] / 1e6,
                        result_ms=lib["d2h"], result_gb_per_s=4.0 * px / lib["d2h"] / 1e6,
                        d2h_ms=lib["d2h"], d2h_gb_per_s=4.0 * px / lib["d2h"] / 1e6)       # (the names of rounds 1-3 for the result leg)
        rec = e2e_leg(False)
        fresh = e2e_leg(True)
        as_runs = trk.stats()["result_as_runs"]
        # the same with the dense result slab written by k_relabel and copied over PCIe (ctk_set_result_transfer 0): the scheme of rounds 1-4
        trk.set_result_transfer(0)
        trk.track(a, thr, op, w, wl["overlap"], wl["persistence"], wl["twosided"])
        dense = e2e_leg(False)
        trk.set_result_transfer(-1)
        out["e2e"] = dict(rec, schema="r4: ms_per_call = results dropped between calls (their memory recycled: ms_per_call_recycled); the quantity "
                                      "rounds 1-3 called ms_per_call (every result a fresh array) is fresh_result_arrays.ms_per_call; d2h_* = result_*",
                          ms_per_call_recycled=rec["ms_per_call"], result_transfer=("run tables, %d block(s) through the write kernel" % (as_runs - 1)) if as_runs else "dense copy",
                          fresh_result_arrays=fresh, dense_copy=dense,
                          note="pageable numpy slab in (plain hipMemcpy at PCIe rate), numpy flag out, includes the %.2f ms device pass.  The result "
                               "crosses PCIe as the pass's run tables (bit mask + first run of every row + value of every run: 1/23 of the int32 "
                               "slab here) and sixteen host threads expand them into the array -- result_gb_per_s is the array's size over that "
                               "leg, bound by the host's DRAM writes, not a PCIe rate.  Headline of this block: results dropped between calls (a "
                               "loop over members; the binding recycles their memory).  fresh_result_arrays: every result kept alive, each call "
                               "writes into pages that do not exist yet.  dense_copy: k_relabel writes the slab in HBM, one DMA into the recycled, "
                               "registered array" % ms_per_step)
    # ensemble members side by side: four handles (own streams and work spaces) driven by four host threads on this one GPU.
    # Not `value` (that is one pass after the other on one handle): what a job with many independent slabs -- BASELINE.json
    # configs[4], 35 members -- gets from the latency-bound middle of one pass running underneath the streaming of another.
    if not args.no_extra and nbytes * 8 < (64 << 30):
        out["concurrent_members"] = concurrent_members(wl, a, thr, op, w, 4, max(args.steps // 2, 4))
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, a, w)
    if a is None:
        # coverage of a device-generated slab without a tracking pass: ctk_check_flag_dev counts the pixels whose flag is nonzero where
        # (double)anom <op> thr[t] is FALSE -- against a "flag" that is nonzero everywhere that is the background (contrack.py:665)
        trk.memset(d_out, 1, px * 4)
        out["config"]["coverage"] = 1.0 - trk.check_flag(d_in, d_out, T, ny, nx, thr, op, 0, 0x01010101)["flag_outside_mask"] / float(px)
    trk.free(d_in)
    trk.free(d_out)
    trk.close()
    # the 0.25 deg grid (what the north_star's >= 50x is stated on) in the same driver-timed process; `value` stays configs[1]
    if args.workload == "era5_1deg_djf30" and not args.no_secondary and not args.no_extra and not args.no_cpu_baseline:
        try:
            out["secondary_025deg"] = secondary_025deg(args.steps, args.warmup)
        except (MemoryError, _native.ContrackHipError) as e:
            out["secondary_025deg"] = dict(error="%s: %s" % (type(e).__name__, e))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
