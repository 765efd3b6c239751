#!/usr/bin/env python
"""Micro-benchmark of one fgt_conv2d layer shape for PMC profiling (rocprofv3 --pmc ... -- python tools/conv_micro.py).

    python tools/conv_micro.py --layer enc8 --tile 128x128 --precision bf16x3 --reps 20
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402

LAYERS = {  # name: (N, H, W, Cin, C1, Cout, groups, k, stride, pad)
    "enc8": (17, 60, 108, 256, 0, 384, 1, 3, 1, 1),          # FGT encoder layer 8 (194.9 GFLOP)
    "enc10": (17, 60, 108, 256, 384, 512, 2, 3, 1, 1),       # grouped two-source (324.9 GFLOP)
    "ffn1": (1, 1, 12240, 512, 0, 1960, 1, 1, 1, 0),         # FFN Linear 512 -> 1960
    "dec3": (17, 120, 216, 64, 0, 64, 1, 3, 1, 1),           # decoder layer3 with x2 upsample (run without here)
    "qkv8": (1, 1, 97920, 512, 0, 1536, 1, 1, 1, 0),          # fused QKV projection of 8 batched windows
    "qkv8_k32": (1, 1, 97920, 32, 0, 1536, 1, 1, 1, 0),       # same output, ONE K-step: prologue + epilogue cost of the tile grid
    "qkv8_k128": (1, 1, 97920, 128, 0, 1536, 1, 1, 1, 0),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="enc8")
    ap.add_argument("--tile", default="128x128")
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--split", default="", choices=["", "planes", "il"], help="feed pre-split inputs (LDS-DMA kernel)")
    a = ap.parse_args()
    N, H, W, C0, C1, Cout, g, k, s, p = LAYERS[a.layer]
    dev = torch.device("cuda:0")
    x = torch.randn(N, H, W, C0, device=dev)
    x1 = torch.randn(N, H, W, C1, device=dev) if C1 else None
    w = torch.randn(Cout, (C0 + C1) // g, k, k, device=dev) * 0.02
    pc = ops.PackedConv(w, torch.zeros(Cout, device=dev), groups=g)
    if a.split:
        x = ops.split(x, interleave=a.split == "il")
        x1 = None if x1 is None else ops.split(x1, interleave=a.split == "il")
    out = None
    for _ in range(3):
        out = ops.conv2d(x, pc, x1=x1, stride=s, pad=p, act="lrelu", tile=a.tile, precision=a.precision, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        ops.conv2d(x, pc, x1=x1, stride=s, pad=p, act="lrelu", tile=a.tile, precision=a.precision, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    fl = 2.0 * out.numel() // Cout * (Cout // g) * pc.K * g
    print(f"{a.layer} tile={a.tile} prec={a.precision} split={a.split or 'no'}: {ms * 1e3:.1f} us/launch, {fl / ms / 1e9:.1f} TFLOP/s algorithmic")


if __name__ == "__main__":
    main()
