#!/usr/bin/env python
"""Register-staged bf16x3 conv (fp32 inputs) vs the LDS-DMA kernel on pre-split inputs, per layer shape and tile.

    python tools/split_sweep.py [--reps 10]        (FGT_CONV_PIPE=0 selects the unpinned schedule of the split kernel)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--diag" in sys.argv:           # before fgt_amd loads the library
    os.environ["FGT_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fgt_amd", "lib", "libfgt_hip_diag.so")
from fgt_amd import ops  # noqa: E402

LAYERS = {  # name: (N, H, W, C0, C1, Cout, groups, k, stride, pad)
    "enc8  256->384 3x3": (17, 60, 108, 256, 0, 384, 1, 3, 1, 1),
    "enc10 640->512 g2": (17, 60, 108, 256, 384, 512, 2, 3, 1, 1),
    "enc12 640->256 g8": (17, 60, 108, 256, 384, 256, 8, 3, 1, 1),
    "e20 enc11 768->384 g4": (20, 60, 108, 256, 512, 384, 4, 3, 1, 1),
    "enc4  64->128 s2": (17, 120, 216, 64, 0, 128, 1, 3, 2, 1),
    "dec   128->128 120x216": (17, 120, 216, 128, 0, 128, 1, 3, 1, 1),
    "dec   64->64 240x432/2": (8, 240, 432, 64, 0, 64, 1, 3, 1, 1),
    "p2v   128->512 7x7s3": (17, 60, 108, 128, 0, 512, 1, 7, 3, 3),
    "ffn1  512->1960": (1, 1, 12240, 512, 0, 1960, 1, 1, 1, 0),
    "ffn2  40->512 7x7s3": (17, 60, 108, 40, 0, 512, 1, 7, 3, 3),
    "qkv   512->1536": (1, 1, 12240, 512, 0, 1536, 1, 1, 1, 0),
    "proj  512->512": (1, 1, 12240, 512, 0, 512, 1, 1, 1, 0),
    "k     768->512": (1, 1, 17340, 768, 0, 512, 1, 1, 1, 0),
    # four equal-length windows batched (ClipRunner.window_batch = 4)
    "b4 ffn1 512->1960": (1, 1, 48960, 512, 0, 1960, 1, 1, 1, 0),
    "b4 ffn2 40->512 7x7s3": (68, 60, 108, 40, 0, 512, 1, 7, 3, 3),
    "b4 qkv  512->1536": (1, 1, 48960, 512, 0, 1536, 1, 1, 1, 0),
    "b4 proj 512->512": (1, 1, 48960, 512, 0, 512, 1, 1, 1, 0),
    "b4 k    768->512": (1, 1, 69360, 768, 0, 512, 1, 1, 1, 0),
    "b8 ffn1 512->1960": (1, 1, 97920, 512, 0, 1960, 1, 1, 1, 0),
    "b8 qkv  512->1536": (1, 1, 97920, 512, 0, 1536, 1, 1, 1, 0),
    "b8 proj 512->512": (1, 1, 97920, 512, 0, 512, 1, 1, 1, 0),
    "b8 k    768->512": (1, 1, 138720, 768, 0, 512, 1, 1, 1, 0),
    "b8 ffn2 40->512 7x7s3": (136, 60, 108, 40, 0, 512, 1, 7, 3, 3),
    "e20 enc8  256->384 3x3": (20, 60, 108, 256, 0, 384, 1, 3, 1, 1),
    "e20 enc10 640->512 g2": (20, 60, 108, 256, 384, 512, 2, 3, 1, 1),
    "e20 p2v   128->512 7x7s3": (20, 60, 108, 128, 0, 512, 1, 7, 3, 3),
    "e20 enc6 128->256": (20, 60, 108, 128, 0, 256, 1, 3, 1, 1),
    "v2p 512->6272 (66 frames)": (1, 1, 47520, 512, 0, 6272, 1, 1, 1, 0),
    # RAFT update block, 32 pairs at 864x480 (60x108 per pair)
    "raft convc2 256->192 3x3": (32, 60, 108, 256, 0, 192, 1, 3, 1, 1),
    "raft fh1 128->256 3x3": (32, 60, 108, 128, 0, 256, 1, 3, 1, 1),
    "raft motion 192+64->128 3x3": (32, 60, 108, 192, 64, 128, 1, 3, 1, 1),
    "raft gru 128+256->128 1x5": (32, 60, 108, 128, 256, 128, 1, (1, 5), 1, (0, 2)),
    "raft gru 128+256->128 5x1": (32, 60, 108, 128, 256, 128, 1, (5, 1), 1, (2, 0)),
    # LAFC, 8 pivots x 3 flows
    "lafc 96->96 3x3 120x216": (24, 120, 216, 96, 0, 96, 1, 3, 1, 1),
    "lafc 192->192 3x3 60x108": (24, 60, 108, 192, 0, 192, 1, 3, 1, 1),
    # nearest x2 upsampling ahead of the conv (" up" in the name): FGT decoder, LAFC decoder
    "dec up 128->128 60x108": (20, 60, 108, 128, 0, 128, 1, 3, 1, 1),
    "dec up 64->64 120x216": (8, 120, 216, 64, 0, 64, 1, 3, 1, 1),
    "lafc up 96+96->48 120x216": (8, 120, 216, 96, 96, 48, 1, 3, 1, 1),
    "lafc up 192+192->96 60x108": (8, 60, 108, 192, 192, 96, 1, 3, 1, 1),
    # 4-channel input convs (FGT flow encoder, RAFT flow encoder) and the same with Cin zero-padded to 8 (split planes need C % 8 == 0)
    "in4 5x5 4->64 240x432": (20, 240, 432, 4, 0, 64, 1, 5, 1, 2),
    "in8 5x5 8->64 240x432": (20, 240, 432, 8, 0, 64, 1, 5, 1, 2),
    "in4 7x7 4->128 60x108": (32, 60, 108, 4, 0, 128, 1, 7, 1, 3),
    "in8 7x7 8->128 60x108": (32, 60, 108, 8, 0, 128, 1, 7, 1, 3),
    # LAFC P3D temporal convs: 3 x 1 over (T = 3, H*W)
    "lafc p3d 96 3x1 T": (8, 3, 25920, 96, 0, 96, 1, (3, 1), 1, (1, 0)),
    "lafc p3d 192 3x1 T": (8, 3, 6480, 192, 0, 192, 1, (3, 1), 1, (1, 0)),
}
TILES = ["128x128", "256x128", "128x128x8", "256x128x16", "256x64x8", "128x128x8ea", "128x128x8eaw"]


def bench(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--layers", default="")
    ap.add_argument("--tiles", default="")
    ap.add_argument("--diag", action="store_true", help="load lib/libfgt_hip_diag.so (fgt_amd.build.build(variant='diag')): the ring / ping-pong / 8-phase / loader-wavefront tiles")
    ap.add_argument("--split-only", action="store_true", help="time only the pre-split (planes) inputs; still checks bit equality with the 128x128 result")
    a = ap.parse_args()
    global TILES
    if a.tiles:
        TILES = a.tiles.split(",")
    dev = torch.device("cuda:0")
    print(f"{'layer':26s} {'GFLOP':>7s} | " + " ".join(f"{t:>18s}" for t in TILES) + "   (TFLOP/s algorithmic: fp32-in / split planes / split interleaved)")
    for name, (N, H, W, C0, C1, Cout, g, k, s, p) in LAYERS.items():
        if a.layers and not any(x in name for x in a.layers.split(",")):
            continue
        up = " up " in name
        x = torch.randn(N, H, W, C0, device=dev)
        x1 = torch.randn(N, H, W, C1, device=dev) if C1 else None
        kh_, kw_ = (k, k) if isinstance(k, int) else k
        w = torch.randn(Cout, (C0 + C1) // g, kh_, kw_, device=dev) * 0.02
        pc = ops.PackedConv(w, torch.zeros(Cout, device=dev), groups=g)
        xs, x1s = ops.split(x), (ops.split(x1) if C1 else None)
        can_il = (C0 // g) % 32 == 0 and (C1 // g) % 32 == 0
        xi, x1i = (ops.split(x, interleave=True), (ops.split(x1, interleave=True) if C1 else None)) if can_il else (None, None)
        out = ops.conv2d(x, pc, x1=x1, stride=s, pad=p, upsample=up, act="lrelu", tile="128x128", precision="bf16x3")
        fl = 2.0 * (out.numel() // Cout) * (Cout // g) * pc.K * g
        cells = []
        for t in TILES:
            o1, o2 = torch.empty_like(out), torch.empty_like(out)
            if t.endswith(("ig", "it", "s3", "s4", "pp", "il", "p8", "p8n", "p8l", "ea", "lw", "xy", "w", "t")) or t.startswith("x2") or a.split_only:        # split inputs only
                ops.conv2d(x, pc, x1=x1, stride=s, pad=p, upsample=up, act="lrelu", tile="128x128", precision="bf16x3", out=o1)
                ms_a = float("inf")
            else:
                ms_a = bench(lambda: ops.conv2d(x, pc, x1=x1, stride=s, pad=p, upsample=up, act="lrelu", tile=t, precision="bf16x3", out=o1), a.reps)
            wide = t.endswith("w")                 # wide LDS image (csrc/conv_wide.hip): interleaved inputs only
            if wide:
                ms_b = float("inf")
                o2.copy_(o1)
            else:
                try:
                    ms_b = bench(lambda: ops.conv2d(xs, pc, x1=x1s, stride=s, pad=p, upsample=up, act="lrelu", tile=t, precision="bf16x3", out=o2), a.reps)
                except RuntimeError as e:                                              # e.g. a tap-reusing tile on a layer that kernel does not serve
                    if "does not serve" not in str(e) and "must be multiples" not in str(e):
                        raise
                    cells.append(f"{fl / ms_a / 1e9:5.0f}/    -/    - ")
                    continue
            ms_c = float("inf")
            if can_il and (wide or not a.split_only):
                o3 = torch.empty_like(out)
                ms_c = bench(lambda: ops.conv2d(xi, pc, x1=x1i, stride=s, pad=p, upsample=up, act="lrelu", tile=t, precision="bf16x3", out=o3), a.reps)
                o2 = o2 if torch.equal(o2, o3) else o2 + 1
            if torch.equal(o1, o2) and torch.equal(o1, out):
                eq = ""
            elif (o2 - out).abs().max().item() <= 2e-5 * out.abs().max().item():
                eq = "~"                                                               # the tap-reusing kernel ("...t" tiles): another accumulation order
            else:
                eq = "!"                                                               # (expected for the timing-only x2* tiles of diagnostic builds)
            cells.append(f"{fl / ms_a / 1e9:5.0f}/{fl / ms_b / 1e9:5.0f}/{fl / ms_c / 1e9:5.0f}{eq:1s}")
        print(f"{name:26s} {fl / 1e9:7.1f} | " + " ".join(cells), flush=True)
        del x, x1, xs, x1s, out


if __name__ == "__main__":
    main()
