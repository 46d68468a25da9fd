#!/usr/bin/env python3
"""tools/summarize_profile.py <gpurun_out/TAG> <profiles/PREFIX> [workload name]
Condenses a tools/profile.sh capture into committed files: PREFIX_kernel_stats.csv (rocprofv3 --stats),
PREFIX_pmc.json / .md (FETCH_SIZE / WRITE_SIZE per kernel, per launch, with the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE counts wide coalesced reads at half their bytes)."""
import collections
import csv
import json
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "era5_1deg_djf30 2707x181x360"
shutil.copy(src + "/trace/r_kernel_stats.csv", dst + "_kernel_stats.csv")


def agg(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return d


f = agg(src + "/pmc_fetch/r_counter_collection.csv", "FETCH_SIZE")
w = agg(src + "/pmc_write/r_counter_collection.csv", "WRITE_SIZE")
stats = {r["Name"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(src + "/trace/r_kernel_stats.csv"))}
out = {}
def full_size(v):
    """the launches of the passes: a kernel is also launched on part of the slab now and then (a first call that repeats a stage, a
    placement check on a window) -- launches below half of the largest are left out of the per-launch mean"""
    m = max(v) if v else 0.0
    keep = [x for x in v if x >= 0.5 * m]
    return keep or v


for k in set(f) | set(w):
    fs, ws = full_size(f.get(k, [0.0])), full_size(w.get(k, [0.0]))
    out[k] = dict(launches=len(fs), fetch_size_kb=sum(fs) / len(fs), write_size_kb=sum(ws) / max(1, len(ws)),
                  fetch_bytes_corrected=2.0 * 1024 * sum(fs) / len(fs), write_bytes=1024.0 * sum(ws) / max(1, len(ws)),
                  avg_ns=float(stats[k]["AverageNs"]) if k in stats else None)
json.dump(dict(note="per launch (launches below half of the largest left out); fetch_bytes_corrected = FETCH_SIZE*1024*2 (gfx950 counts wide coalesced reads at half), "
                    "write_bytes = WRITE_SIZE*1024 (uncalibrated); workload " + workload,
               kernels=out), open(dst + "_pmc.json", "w"), indent=1, sort_keys=True)
with open(dst + "_pmc.md", "w") as fh:
    fh.write("| kernel | launches | avg us | FETCH_SIZE KB | x2 corrected MB | WRITE_SIZE KB | MB |\n|---|---|---|---|---|---|---|\n")
    for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["fetch_bytes_corrected"] + kv[1]["write_bytes"])):
        fh.write("| %s | %d | %s | %.1f | %.1f | %.1f | %.1f |\n" % (
            k, v["launches"], "%.1f" % (v["avg_ns"] / 1e3) if v["avg_ns"] else "-", v["fetch_size_kb"], v["fetch_bytes_corrected"] / 1e6,
            v["write_size_kb"], v["write_bytes"] / 1e6))
print("wrote", dst + "_kernel_stats.csv", dst + "_pmc.json", dst + "_pmc.md")
