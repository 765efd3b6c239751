#!/usr/bin/env python
"""Run-to-run reproducibility of single tap-kernel launches under GPU sharing (start two or three of these side by side): every geometry x tile is
launched `--reps` times, every output compared bit for bit with the first.  FGT_TAPS_WIDE (0 / 1 / 2) and FGT_HIP_LIB select the image and the build.
    python tools/layer_race_check.py [--reps 150]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=150)
a = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.DEFAULT_CONV_PRECISION = "bf16x3"
g = torch.Generator().manual_seed(0)
LAYERS = [  # name, N, H, W, C0, C1, Cout, groups
    ("enc4 256->384", 20, 60, 108, 256, 0, 384, 1),
    ("enc5 256+384->512 g2", 20, 60, 108, 256, 384, 512, 2),
    ("enc6 256+512->384 g4", 20, 60, 108, 256, 512, 384, 4),
    ("enc7 256+384->256 g4 (merged)", 20, 60, 108, 256, 384, 256, 4),
    ("enc8 256+256->128", 20, 60, 108, 256, 256, 128, 1),
    ("dec 128->128 120x216", 20, 120, 216, 128, 0, 128, 1),
    ("token grid 512->384 (W = 36)", 136, 20, 36, 512, 0, 384, 1),
]
for name, N, H, W, C0, C1, Cout, G in LAYERS:
    x0 = ops.split(torch.randn(N, H, W, C0, generator=g).to(dev), interleave=True)
    x1 = ops.split(torch.randn(N, H, W, C1, generator=g).to(dev), interleave=True) if C1 else None
    Cg = (C0 + C1) // G
    pc = ops.PackedConv((torch.randn(Cout, Cg, 3, 3, generator=g) / math.sqrt(9 * Cg)).to(dev), None, groups=G)
    for tile in ("128x128it", "256x128it", "256x256it", "128x128x8t"):
        try:
            ref = ops.conv2d(x0, pc, x1=x1, pad=1, act="lrelu", tile=tile)
        except RuntimeError:
            continue
        torch.cuda.synchronize()
        bad = 0
        for _ in range(a.reps):
            out = ops.conv2d(x0, pc, x1=x1, pad=1, act="lrelu", tile=tile)
            if not torch.equal(out, ref):
                bad += 1
        print(f"pid {os.getpid()} {name:34s} {tile:11s}: {bad} of {a.reps} launches differ from the first", flush=True)
