#!/usr/bin/env python
"""What the output FORM of a conv costs: one short-K layer (the flow encoder's 5x5 4 -> 64 at 240x432, 20 frames: all epilogue) and one K = 512 GEMM
(97 920 x 512 -> 512), each with fp32 / split-planes / split-interleaved / both outputs, same tile.    python tools/epilogue_micro.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.DEFAULT_CONV_PRECISION = "bf16x3"
g = torch.Generator().manual_seed(0)


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


x = torch.randn(20, 240, 432, 4, generator=g).to(dev)
pc = ops.PackedConv((torch.randn(64, 4, 5, 5, generator=g) * 0.1).to(dev), torch.zeros(64, device=dev))
a = torch.randn(97920, 512, generator=g).to(dev)
asp = ops.split(a, interleave=True)
pl = ops.PackedConv((torch.randn(512, 512, generator=g) * 0.04).to(dev), torch.zeros(512, device=dev))
for name, run, rows, cout in (("5x5 4->64, 240x432 x 20", lambda **kw: ops.conv2d(x, pc, pad=2, pad_mode="replicate", act="lrelu", **kw), 20 * 240 * 432, 64),
                              ("GEMM 97920 x 512 -> 512", lambda **kw: ops.linear(asp, pl, **kw), 97920, 512)):
    for tile in ("128x128", "64x64", "128x64", "256x128"):
        row = []
        for form, kw in (("fp32", {}), ("planes", dict(out_split="only")), ("interleaved", dict(out_split="only", out_il=True)), ("both", dict(out_split="both", out_il=True))):
            try:
                if form in ("planes", "interleaved") and "GEMM" in name:
                    out_s = ops.Split.empty((rows, cout), dev, interleaved=form == "interleaved", h=False)
                    ms = timed(lambda: run(tile=tile, out_split="only", out_s=out_s))
                elif form == "both" and "GEMM" in name:
                    out_s = ops.Split.empty((rows, cout), dev, interleaved=True, h=False)
                    ms = timed(lambda: run(tile=tile, out_split="both", out_s=out_s))
                else:
                    ms = timed(lambda: run(tile=tile, **kw))
            except RuntimeError as e:
                row.append(f"{form}: n/a")
                continue
            nbytes = rows * cout * (8 if form == "both" else 4)
            row.append(f"{form}: {ms:6.3f} ms ({nbytes / ms / 1e6:6.0f} GB/s out)")
        print(f"{name:26s} {tile:8s} " + " | ".join(row))
