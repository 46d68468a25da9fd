"""Kernel timeline of one bench pass from a rocprofv3 --kernel-trace CSV: python tools/timeline.py <dir> [pass]
(pass: index among the k_threshold launches, default -6 = the last TIMED pass of bench.py, which appends 5 passes with
events around every kernel group)"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_threshold" in r["Kernel_Name"]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -6
idx = starts[which]
end = starts[which + 1] if which + 1 < 0 or (which >= 0 and which + 1 < len(starts)) else len(rows)
t0 = int(rows[idx]["Start_Timestamp"])
prev_end = t0
for r in rows[idx:end]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:60]))
    prev_end = e
