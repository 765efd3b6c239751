#!/usr/bin/env python
"""Timing-only ablations of the split-input attention tile loop (diagnostic build -DFGT_ATTN_ABLATE, never the product library; results of the
ablated runs are wrong on purpose).  FGT_ATTN_ABLATE bits: 1 no QK^T MFMAs (+ K fragment reads), 2 no exp2 in the softmax, 4 no PV product (+ V reads,
P conversion), 8 no LDS-DMA after the prologue, 16 no barrier.

    python tools/attn_ablate.py --build                      (here: cross-compiles fgt_amd/lib/libfgt_hip_ablate.so)
    python tools/attn_ablate.py                              (on the MI355X)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "fgt_amd", "lib", "libfgt_hip_ablate.so")
CASES = [(0, "full"), (1, "no QK"), (2, "no exp"), (4, "no PV"), (5, "no QK, no PV"), (7, "no QK, exp, PV"), (8, "no DMA in the loop"), (16, "no barrier"),
         (24, "no DMA, no barrier"), (13, "no QK, PV, DMA"), (31, "nothing but the loop skeleton + max / sum VALU")]


def run():
    os.environ["FGT_HIP_LIB"] = LIB                          # before fgt_amd loads the library
    import torch
    from fgt_amd import ops
    dev = torch.device("cuda:0")
    for fmt in ("f16", "bf16x3"):
        for t, b in ((17, 8),):
            qkv = ops.split(torch.randn(b * t * 720, 1536, device=dev), h=(fmt == "f16"))
            fn = lambda: ops.attention_temporal(qkv, b, t, 20, 36, 4, 2, 512)
            base = None
            for bits, name in CASES:
                os.environ["FGT_ATTN_ABLATE"] = str(bits)    # read by the library at every launch (getenv)
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 8 * 1e3
                base = base or us
                print(f"{fmt:7s} b={b} t={t}  {name:48s} {us:9.1f} us  {us / base:5.2f}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    a = ap.parse_args()
    if a.build:
        from fgt_amd import build as B
        print(B.build(variant="ablate", extra_flags=["-DFGT_ATTN_ABLATE"], swap={"attention_split.hip": "diag/attention_split_trace.hip"}, verbose=False))
        return
    run()


if __name__ == "__main__":
    main()
