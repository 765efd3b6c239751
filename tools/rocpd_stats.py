#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (sqlite) kernel trace as a --stats style table (per-kernel calls, total/avg/min/max ns, %).

    python tools/rocpd_stats.py gpurun_out/prof/fgt_results.db [> profiles/rNN_kernel_stats.csv]
"""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(f"select {name_col}, start, end from kernels"))
    agg = {}
    for name, s, e in rows:
        short = name.replace("(anonymous namespace)::", "")
        short = re.sub(r"^void ", "", short)
        short = re.sub(r"\((?:[^()]|\([^()]*\))*\)\s*(\[clone[^\]]*\])?$", "", short)   # drop the argument list
        a = agg.setdefault(short, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'"{k}",{a[0]},{a[1]},{a[1] / a[0]:.0f},{100.0 * a[1] / tot:.2f},{a[2]},{a[3]}')
    print(f'"TOTAL",{len(rows)},{tot},,,,')


if __name__ == "__main__":
    main(sys.argv[1])
