#!/usr/bin/env python
"""Effective clock and matrix-pipe occupancy per kernel from one rocprofv3 --pmc pass over bench.py.

    (cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
        --kernel-trace --output-format csv -d <dir> -o pmc -- python bench.py --steps 1 --warmup 0 --no-prof ...)
    python tools/pmc_clock.py <dir>

effective clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (MI355X_MICROARCH.md "DVFS give-back"); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).  The nominal bf16 peak (2.5 PF) assumes 2.4 GHz: a kernel at clock f can reach at most f / 2.4 of it.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main(d):
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    per = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    agg = defaultdict(lambda: defaultdict(float))
    for did, c in per.items():
        if did not in dur:
            continue
        ns, name = dur[did]
        short = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
        a = agg[short]
        a["ns"] += ns
        a["n"] += 1
        for k, v in c.items():
            a[k] += v
    tot = sum(a["ns"] for a in agg.values())
    print(f"{'kernel':58s} {'calls':>5s} {'ms':>8s} {'%':>5s} {'GHz':>5s} {'MFMA busy':>9s} {'wave: active':>12s} {'wait':>5s} {'issue-stall':>11s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"])[:14]:
        cyc = a["GRBM_GUI_ACTIVE"] / 8.0
        ghz = cyc / a["ns"] if a["ns"] else 0
        mf = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) if cyc else 0
        wc = a["SQ_WAVE_CYCLES"] or 1
        print(f"{k[:58]:58s} {int(a['n']):5d} {a['ns'] / 1e6:8.2f} {100 * a['ns'] / tot:5.1f} {ghz:5.2f} {100 * mf:8.1f}% {100 * a['SQ_ACTIVE_INST_ANY'] / wc:11.0f}% {100 * a['SQ_WAIT_ANY'] / wc:4.0f}% {100 * a['SQ_WAIT_INST_ANY'] / wc:10.0f}%")


if __name__ == "__main__":
    main(sys.argv[1])
