"""Kernel timeline of the last bench pass from a rocprofv3 --kernel-trace CSV: python tools/timeline.py <dir>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_threshold" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"])
prev_end = t0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:60]))
    prev_end = e
