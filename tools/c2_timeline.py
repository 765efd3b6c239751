#!/usr/bin/env python
"""Kernel timeline of the LAST replay in a rocprofv3 --kernel-trace CSV of tools/c2_module.py: start / end (us, relative), overlap with the previous kernel."""
import csv
import glob
import sys

path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = 0
busy = 0
for r in last:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f} us  {'OVERLAPS prev by %.1f' % (prev_end - s) if s < prev_end else '':22s} {r['Kernel_Name'][:90]}")
    prev_end = max(prev_end, e)
    busy += e - s
print(f"span {prev_end:.1f} us, sum of kernel times {busy:.1f} us")
