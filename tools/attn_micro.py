#!/usr/bin/env python
"""Micro-benchmark of the temporal attention kernel at the bench shape (t frames, 20x36 tokens, 4 heads, 2x2 zones)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--t", type=int, default=17)
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--spatial", action="store_true")
    ap.add_argument("--b", type=int, default=1, help="windows per call (the bench batches 8 equal-length windows)")
    ap.add_argument("--split", action="store_true", help="pre-split inputs (csrc/attention_split.hip: LDS-DMA K / V tiles)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.spatial:
        bt, h, w, nh, nw = a.t, 20, 36, 24, 40
        q, k, v = (torch.randn(bt * nh * nw, 512, device=dev) for _ in range(3))
        kg, vg = torch.randn(bt * 60, 512, device=dev), torch.randn(bt * 60, 512, device=dev)
        if a.split:
            q, k, v, kg, vg = (ops.split(x) for x in (q, k, v, kg, vg))
        fn = lambda: ops.attention_spatial(q, k, v, kg, vg, bt, h, w, nh, nw, 4, 8, 60, precision=a.precision)
        flops = 4.0 * bt * 15 * 4 * 64 * 124 * 128
    else:
        qkv = torch.randn(a.b * a.t * 720, 1536, device=dev)
        if a.split:
            qkv = ops.split(qkv)
        fn = lambda: ops.attention_temporal(qkv, a.b, a.t, 20, 36, 4, 2, 512, precision=a.precision)
        L = a.t * 180
        flops = 4.0 * 16 * a.b * L * L * 128
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print(f"attention {'spatial' if a.spatial else 'temporal'} b={a.b} t={a.t} {a.precision}{' split-in' if a.split else ''}: {ms * 1e3:.1f} us, {flops / ms / 1e9:.1f} TFLOP/s algorithmic")


if __name__ == "__main__":
    main()
