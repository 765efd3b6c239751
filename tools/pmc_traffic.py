#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py into per-launch HBM traffic of the MFMA kernels AND (round 5) of the
bandwidth-bound kernels (LayerNorm, fold, small-Cout convs, depthwise pools, the pointwise family; warp and correlation lookup from a pass over
tools/hbm_micro.py, which runs them at the bench's shapes).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <precision> <out.json> [<command note>] [--merge kind,kind,...] [--alg alg.json]

--merge: add / replace only the listed kinds in <out.json>[precision] (the micro command's pass) instead of rewriting the entry;
--alg: {kind: {"algorithmic_bytes_per_launch": ..}} written by tools/hbm_micro.py, stored beside the counter figures.
Per kernel KIND (the grouping bench.py's `rooflines` use: conv = conv_split + conv_igemm, attn_temporal, attn_spatial, layernorm, ...) and per
kernel TEMPLATE instance.  The profiled command must run with FGT_TUNING_FILE pre-seeded so that no autotuner candidate launch
is in the trace: every launch counted is a launch of a clip pass.  FETCH_SIZE x2: gfx950 tallies 128-B requests at 64 B
(MI355X_MICROARCH.md §HBM); both counters are in KiB.
"""
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict


def kind_of(name):
    if "conv_split_kernel" in name or "conv_igemm_kernel" in name or "conv_f16" in name or "conv_wide" in name or "conv_taps" in name:
        return "conv"
    if "attn_" in name:
        # spatial windows run the 2-wavefront instances (64 queries), temporal zones the 4- / 8-wavefront ones
        return "attn_spatial" if re.search(r"attn_(bf16x3_|split_)?kernel<2", name) else "attn_temporal"
    # HBM-bound kernels (bench.py HBM_KERNELS / bench_stages' warp, corr_lookup): the kinds of fgt_prof_* (include/fgt_hip.h FGT_PROF_*)
    for kind, pats in HBM_KINDS.items():
        if any(p_ in name for p_ in pats):
            return kind
    return None


HBM_KINDS = {"layernorm": ("layernorm_kernel",), "fold": ("fold_kernel",), "conv_small": ("conv3x3_tiled_kernel", "conv_direct_kernel"),
             "dw_pool": ("dw_pool4_kernel", "dw_pool_kernel"), "warp": ("warp_kernel", "warp_c2x2_kernel"), "corr_lookup": ("corr_lookup_kernel",),
             "pointwise": ("gather_rows_kernel", "dw3x3_res_kernel", "split_kernel", "split2_kernel", "pad_tokens_kernel", "pack_frames_kernel",
                           "nchw_to_nhwc_kernel", "nhwc_to_nchw_kernel", "axpby_kernel")}



def collect(d, counter):
    per = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r.get("Kernel_Name", "")
            if kind_of(name) is None:
                continue
            short = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
            a = per[short]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return per


def main():
    argv = list(sys.argv[1:])
    merge, alg = None, {}
    if "--merge" in argv:
        i = argv.index("--merge")
        merge = set(argv[i + 1].split(","))
        del argv[i:i + 2]
    if "--alg" in argv:
        i = argv.index("--alg")
        alg = json.load(open(argv[i + 1]))
        del argv[i:i + 2]
    fd, wd, prec, out = argv[:4]
    note = argv[4] if len(argv) > 4 else "bench.py --steps 1 --warmup 0 --no-prof --no-cpu-baseline --no-fp32-exact --no-f16 --no-c4 (prepare pass + 1 step, tiles pre-seeded)"
    F, W = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
    kinds = defaultdict(lambda: {"launches": 0, "fetch_KiB": 0.0, "write_KiB": 0.0})
    templates = {}
    for name in sorted(set(F) | set(W)):
        nf, f = F.get(name, [0, 0.0])
        nw, w = W.get(name, [0, 0.0])
        n = max(nf, nw, 1)
        templates[name] = {"launches": nf, "hbm_bytes_per_launch": round((2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024)}
        k = kinds[kind_of(name)]
        k["launches"] += nf
        k["fetch_KiB"] += f
        k["write_KiB"] += w * (nf / max(nw, 1))
    res = json.load(open(out)) if os.path.exists(out) else {}
    try:
        head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        head = ""
    if not head:        # the GPU box gets a snapshot without .git: tools/gpu.sh writes the head of the snapshot into .git_head
        hf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".git_head")
        head = open(hf).read().strip() if os.path.exists(hf) else ""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fgt_amd.build import csrc_hash
    entry = {"command": note, "git_head": head, "csrc_hash": csrc_hash(), "templates": templates,
             "note": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md §HBM) + WRITE_SIZE, KiB -> bytes, per launch"}
    for k, v in kinds.items():
        n = max(v["launches"], 1)
        entry[k] = {"launches": v["launches"], "fetch_KiB_per_launch_raw": v["fetch_KiB"] / n, "write_KiB_per_launch": v["write_KiB"] / n,
                    "hbm_bytes_per_launch": round((2.0 * v["fetch_KiB"] + v["write_KiB"]) / n * 1024)}
        if k in alg:
            entry[k].update(alg[k])
            if alg[k].get("algorithmic_bytes_per_launch"):
                entry[k]["traffic_over_algorithmic"] = round(entry[k]["hbm_bytes_per_launch"] / alg[k]["algorithmic_bytes_per_launch"], 3)
    if merge is not None:
        cur = res.get(prec, {})
        for k in kinds:
            if k in merge:
                cur[k] = dict(entry[k], command=note, git_head=head, csrc_hash=csrc_hash())
        cur.setdefault("templates", {}).update({n_: t for n_, t in templates.items() if kind_of(n_) in merge})
        res[prec] = cur
        json.dump(res, open(out, "w"), indent=1)
        print({k: cur[k] for k in kinds if k in merge})
        return
    res[prec] = entry
    json.dump(res, open(out, "w"), indent=1)
    print({k: entry[k] for k in kinds})


if __name__ == "__main__":
    main()
