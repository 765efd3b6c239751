#!/usr/bin/env python
"""Yardstick (tools only, never product): hipBLASLt's bf16 GEMM (torch.matmul) on the step's real GEMM shapes next to the bf16x3 kernels of
libfgt_hip.so on the same shapes (VERDICT r3 next #2a).  Random operands; TFLOP/s ISSUED (bf16x3: algorithmic x 3 MFMA passes).

    python tools/gemm_yardstick.py [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402

SHAPES = [(97920, 512, 1536, "qkv"), (97920, 512, 1960, "ffn1"), (97920, 512, 512, "proj"), (97920, 1960, 512, "ffn2 as GEMM"),
          (98304, 4096, 512, "long-K 4096"), (8192, 8192, 8192, "8192^3")]


def bench(fn, reps):
    for _ in range(3):
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
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    print(f"{'shape (M x K x N)':34s} {'hipBLASLt bf16':>16s} {'bf16x3 auto tile':>18s} {'issued (x3)':>12s}   TFLOP/s")
    for M, K, N, name in SHAPES:
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.02
        a16, w16 = A.bfloat16(), W.bfloat16().t().contiguous()
        fl = 2.0 * M * K * N
        ms_lt = bench(lambda: torch.matmul(a16, w16), a.reps)
        cell = "-"
        if K % 32 == 0:
            pc = ops.PackedConv(W.view(N, K, 1, 1).contiguous(), torch.zeros(N, device=dev))
            xs = ops.split(A.view(1, 1, M, K), interleave=True)
            out = torch.empty(1, 1, M, N, device=dev)
            ms = bench(lambda: ops.conv2d(xs, pc, precision="bf16x3", out=out), a.reps)
            cell = f"{fl / ms / 1e9:18.1f} {3 * fl / ms / 1e9:12.1f}"
        print(f"{name + ' ' + str((M, K, N)):34s} {fl / ms_lt / 1e9:16.1f} {cell}", flush=True)
        del A, W, a16, w16


if __name__ == "__main__":
    main()
