#!/usr/bin/env python
"""Per-kernel sums of arbitrary counters from rocprofv3 --pmc passes (--output-format csv).
    python tools/pmc_any.py <dir> [<dir> ...] [--top N] [--match substr]
Prints, per kernel name (launches summed), every counter found, sorted by the first counter."""
import csv
import glob
import sys
from collections import defaultdict

args = sys.argv[1:]
top, match = 12, None
dirs = []
i = 0
while i < len(args):
    if args[i] == "--top":
        top = int(args[i + 1]); i += 2
    elif args[i] == "--match":
        match = args[i + 1]; i += 2
    else:
        dirs.append(args[i]); i += 1
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
names = []
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k, c = r["Kernel_Name"], r["Counter_Name"]
            if match and match not in k:
                continue
            if c not in names:
                names.append(c)
            acc[k][c] += float(r["Counter_Value"])
            cnt[k][c] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get(names[0], 0.0))[:top] if names else []
for k, v in rows:
    print(k[:110])
    print("    launches", max(cnt[k].values()), " ".join(f"{c}={v.get(c, 0.0):.4g}" for c in names))
