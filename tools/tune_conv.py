#!/usr/bin/env python
"""Per-layer timing of every distinct fgt_conv2d call of one FGT window forward (t frames at 240x432) under each tile
configuration.  Run on the GPU box; writes gpurun_out/tune_conv.json and prints a table (TFLOP/s per layer x tile).

    python tools/tune_conv.py [--t 17] [--tiles 128x128,128x64,64x64,128x32,256x128]
"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402
from fgt_amd.fgt_model import DEFAULT_CONFIG, Model  # noqa: E402
from fgt_amd.synth import synth_clip, synth_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--t", type=int, default=17)
    ap.add_argument("--tiles", default="auto,128x128,128x64,64x64,128x32,256x128")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    m = Model(dict(DEFAULT_CONFIG)).eval()
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0), strict=True)
    m = m.to(dev)
    fr, fl, ms = synth_clip(args.t, 240, 432, device=dev)
    calls = []
    real = ops.conv2d

    def spy(x, pc, **kw):
        out = real(x, pc, **kw)
        calls.append((x, pc, dict(kw), out))
        return out

    ops.conv2d = spy
    import fgt_amd.fgt_model as fm
    fm.ops.conv2d = spy
    m((fr * 2 - 1) * (1 - ms), fl, ms)
    ops.conv2d = real
    fm.ops.conv2d = real
    torch.cuda.synchronize()
    uniq = {}
    for x, pc, kw, out in calls:
        unwrap = lambda t: t.hi if isinstance(t, ops.Split) else t
        x4 = ops._as_map(unwrap(x))[0]
        x1 = kw.get("x1")
        c1 = 0 if x1 is None else ops._as_map(unwrap(x1))[4]
        key = (tuple(x4.shape), "split" if isinstance(x, ops.Split) else "fp32", c1, pc.Cout, pc.groups, pc.kh, pc.kw, str(kw.get("stride", 1)), str(kw.get("pad", 0)), bool(kw.get("upsample")), kw.get("epi"))
        if key not in uniq:
            uniq[key] = [x, pc, kw, out, 0]
        uniq[key][4] += 1
    tiles = args.tiles.split(",")
    rows = []
    for key, (x, pc, kw, out, count) in uniq.items():
        osp = kw.get("out_split")
        o32, o_s = (out if osp == "both" else ((None, out) if osp == "only" else (out, None)))
        M = math.prod((o32 if o32 is not None else o_s).shape) // pc.Cout
        flops = 2.0 * M * (pc.Cout // pc.groups) * pc.K * pc.groups
        row = {"in": list(key[0]) + [key[1]], "C1": key[2], "Cout": pc.Cout, "groups": pc.groups, "k": [pc.kh, pc.kw], "stride": key[7], "pad": key[8],
               "upsample": key[9], "epi": key[10], "calls": count, "gflop": flops / 1e9, "tf": {}}
        kw2 = dict(kw)
        kw2["out"], kw2["out_s"] = o32, o_s
        for tile in tiles:
            kw2["tile"] = tile
            try:
                for _ in range(2):
                    real(x, pc, **kw2)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    real(x, pc, **kw2)
                e1.record()
                torch.cuda.synchronize()
                ms_ = e0.elapsed_time(e1) / args.reps
                row["tf"][tile] = round(flops / ms_ / 1e9, 2)
            except Exception as ex:  # noqa
                row["tf"][tile] = None
        rows.append(row)
    rows.sort(key=lambda r: -r["gflop"] * r["calls"])
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open("gpurun_out/tune_conv.json", "w"), indent=1)
    tot = sum(r["gflop"] * r["calls"] for r in rows)
    print(f"t={args.t}: {len(calls)} conv2d launches, {len(rows)} distinct, {tot / 1e3:.2f} TFLOP total")
    print("%-36s %4s %5s %3s %6s %5s %8s | " % ("input", "C1", "Cout", "g", "k", "calls", "GFLOP") + " ".join("%8s" % t for t in tiles))
    for r in rows:
        print("%-36s %4d %5d %3d %6s %5d %8.2f | " % (str(r["in"]), r["C1"], r["Cout"], r["groups"], "%dx%d" % tuple(r["k"]), r["calls"], r["gflop"]) +
              " ".join("%8s" % ("-" if r["tf"][t] is None else "%.1f" % r["tf"][t]) for t in tiles))


if __name__ == "__main__":
    main()
