#!/usr/bin/env python
"""What do the GRU convs' epilogue operands cost?  RAFT's z || r launch (bias map + aux1 + two output forms) and q launch (bias map + GRU combine with two
operands, fp32 + split outputs) against the SAME convolution with a plain epilogue, per tap tile, at 32 pairs of 864x480.    python tools/gru_epilogue_cost.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402
from fgt_amd.ops import PackedConv  # noqa: E402

ops.DEFAULT_CONV_PRECISION = "bf16x3"
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
B, H, W = 32, 60, 108
rows = B * H * W
g = torch.Generator().manual_seed(0)
h = torch.randn(rows, 128, generator=g).to(dev)
z = torch.rand(rows, 128, generator=g).to(dev)
m = torch.randn(rows, 128, generator=g).to(dev)
hs, ms = ops.split(h, h=False), ops.split(m, h=False)
v4 = lambda s: s.view(B, H, W, 128)
bm = torch.randn(rows, 256, generator=g).to(dev)
bq = bm[:, :128].contiguous()
rh = ops.Split.empty((rows, 128), dev, h=False)
hn = ops.Split.empty((rows, 128), dev, h=False)


def timed(fn, reps=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


for k, pad in (((1, 5), (0, 2)), ((5, 1), (2, 0))):
    wzr, wq = (torch.randn(256, 256, *k, generator=g) * 0.03).to(dev), (torch.randn(128, 256, *k, generator=g) * 0.03).to(dev)
    pzr, pq = PackedConv(wzr, torch.randn(256, generator=g).to(dev)), PackedConv(wq, torch.randn(128, generator=g).to(dev))
    gf = 2.0 * rows * 128 * 256 * 5 / 1e9
    print(f"== k {k}")
    for t in ops.TAPS_CANDIDATES:
        res = []
        for name, n, fn in (("zr plain f32", 2, lambda: ops.conv2d(v4(hs), pzr, x1=v4(ms), pad=pad, act="sigmoid", tile=t)),
                            ("zr plain split", 2, lambda: ops.conv2d(v4(hs), pzr, x1=v4(ms), pad=pad, act="sigmoid", out_split="only", out_h=False, tile=t)),
                            ("zr bias map", 2, lambda: ops.conv2d(v4(hs), pzr, x1=v4(ms), bias_map=bm, pad=pad, act="sigmoid", tile=t)),
                            ("zr full", 2, lambda: ops.conv2d(v4(hs), pzr, x1=v4(ms), bias_map=bm, pad=pad, act="sigmoid", epi="mul", aux1=h, out_split="both", out_s=rh, dual=True, tile=t)),
                            ("q plain f32", 1, lambda: ops.conv2d(v4(hs), pq, x1=v4(ms), pad=pad, act="tanh", tile=t)),
                            ("q full", 1, lambda: ops.conv2d(v4(hs), pq, x1=v4(ms), bias_map=bq, pad=pad, act="tanh", epi="gru", aux1=z, aux2=h, out_split="both", out_s=hn, tile=t))):
            try:
                ms_ = timed(fn)
                res.append(f"{name} {ms_:.3f} ms {gf * n / ms_:.0f} TF")
            except RuntimeError:
                res.append(f"{name} n/a")
        print(f"  {t:12s} " + " | ".join(res))
