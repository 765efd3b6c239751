#!/usr/bin/env python
"""fp16 conv kernel (csrc/conv_f16.hip): per layer shape, TFLOP/s (algorithmic) of each tile on the narrow LDS image (two 64-byte half rows per
pixel, 16-row x 64-byte LDS-DMA pieces) and on the wide one (`...w`: 128-byte rows, 8-row x 128-byte pieces = full cache lines).

    python tools/f16_sweep.py [--reps 10] [--tiles 128x128x8,128x128,...]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402
from split_sweep import LAYERS, bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tiles", default="128x128x8,128x128,128x128x8ea,128x128ea,256x128,256x128ea,256x128x16")
    ap.add_argument("--layers", default="")
    a = ap.parse_args()
    tiles = a.tiles.split(",")
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    names = [n for n in LAYERS if not a.layers or any(k in n for k in a.layers.split(","))]
    print(f"{'layer':28s} " + " ".join(f"{t:>13s} {'w':>6s}" for t in tiles))
    for name in names:
        N, H, W, C0, C1, Cout, g, k, s, p = LAYERS[name]
        gen = torch.Generator().manual_seed(0)
        x0 = ops.split(torch.randn(N, H, W, C0, generator=gen).to(dev), h=True)
        x1 = ops.split(torch.randn(N, H, W, C1, generator=gen).to(dev), h=True) if C1 else None
        w = torch.randn(Cout, (C0 + C1) // g, k, k, generator=gen) * 0.02
        pc = ops.PackedConv(w.to(dev), torch.zeros(Cout, device=dev), groups=g)
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        fl = 2.0 * N * Ho * Wo * Cout * pc.k_alg
        out = torch.empty(N, Ho, Wo, Cout, device=dev)
        cells = []
        for t in tiles:
            for suffix in ("", "w"):
                try:
                    ms = bench(lambda: ops.conv2d(x0, pc, x1=x1, stride=s, pad=p, act="lrelu", tile=t + suffix, out=out), a.reps)
                    cells.append(fl / ms / 1e9)
                except RuntimeError:
                    cells.append(float("nan"))
        print(f"{name:28s} " + " ".join(f"{cells[2 * i]:13.1f} {cells[2 * i + 1]:6.1f}" for i in range(len(tiles))), flush=True)


if __name__ == "__main__":
    main()
