#!/usr/bin/env python
"""RAFT's GRU z / r convs (update.py:46-49, 53-56) as two Cout = 128 launches vs ONE two-headed Cout = 256 launch (fgt_conv_desc.dual_n0), per tile:
ms per launch at 32 pairs of 864x480 (207 360 rows), bias maps and epilogue operands as in raft_model.iterate.    python tools/zr_micro.py"""
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
m = torch.randn(rows, 128, generator=g).to(dev)
hs, ms = ops.split(h, h=False), ops.split(m, h=False)
v4 = lambda s: s.view(B, H, W, 128)
bm = torch.randn(rows, 256, generator=g).to(dev)
bz, br = bm[:, :128].contiguous(), bm[:, 128:].contiguous()
rh = ops.Split.empty((rows, 128), dev, h=False)


def timed(fn, reps=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


for k, pad in (((1, 5), (0, 2)), ((5, 1), (2, 0))):
    wz, wr = (torch.randn(128, 256, *k, generator=g) * 0.03).to(dev), (torch.randn(128, 256, *k, generator=g) * 0.03).to(dev)
    pz, pr, pzr = PackedConv(wz, None), PackedConv(wr, None), PackedConv(torch.cat([wz, wr], 0), None)
    gf = 2.0 * rows * 128 * 256 * 5 / 1e9
    print(f"== k {k}: {gf:.1f} GFLOP per Cout = 128 conv")
    for t in ops.TAPS_CANDIDATES:
        res = []
        for name, fn in (("z", lambda: ops.conv2d(v4(hs), pz, x1=v4(ms), bias_map=bz, pad=pad, act="sigmoid", tile=t)),
                         ("r", lambda: ops.conv2d(v4(hs), pr, x1=v4(ms), bias_map=br, pad=pad, act="sigmoid", epi="mul", aux1=h, out_split="only", out_s=rh, tile=t)),
                         ("zr", lambda: ops.conv2d(v4(hs), pzr, x1=v4(ms), bias_map=bm, pad=pad, act="sigmoid", epi="mul", aux1=h, out_split="both", out_s=rh, dual=True, tile=t))):
            try:
                ms_ = timed(fn)
                res.append(f"{name} {ms_:.3f} ms {gf * (2 if name == 'zr' else 1) / ms_:.0f} TF")
            except RuntimeError as e:
                res.append(f"{name} n/a")
        print(f"  {t:12s} " + " | ".join(res))
