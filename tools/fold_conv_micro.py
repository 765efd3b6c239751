#!/usr/bin/env python
"""The FFN's fold convolution (Linear 512 -> 1960 + nn.Fold(7, 3, 3) / fold(ones) + ReLU as ONE 3x3 token-grid conv, fgt_conv_desc.ps_r) at the
bench's launch shape (136 frames x 20 x 36 tokens), per tile and tile order, next to the formulation it replaces (K = 512 GEMM + fgt_fold):
    python tools/fold_conv_micro.py [--frames 136] [--v2p]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import fgt_model as M, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=136)
ap.add_argument("--v2p", action="store_true", help="Vec2Patch's shape (128 channels per pixel, + residual) instead of the FFN's (40, / count, ReLU)")
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.DEFAULT_CONV_PRECISION = "bf16x3"
N, th, tw, Hf, Wf, cin, k, s, p = a.frames, 20, 36, 60, 108, 512, 7, 3, 3
cc = 128 if a.v2p else 40
g = torch.Generator().manual_seed(0)
x = torch.randn(N * th * tw, cin, generator=g).to(dev)
w = (torch.randn(cc * k * k, cin, generator=g) / math.sqrt(cin)).to(dev)
b = torch.randn(cc * k * k, generator=g).to(dev)
res = torch.randn(N, Hf, Wf, cc, generator=g).to(dev) if a.v2p else None
g0, cout, _ = M.fold_conv_layout(cc, s)
pc = ops.PackedConv(M.fold_conv_weight(w, cc, k, s), None)
pc.k_alg = cin
off, sc = M.fold_conv_tables(b, cc, k, s, th, tw, not a.v2p)
off, sc = off.to(dev), (None if sc is None else sc.to(dev))
xs = ops.split(x, interleave=True)
x4 = xs.view(N, th, tw, cin)
flops = 2.0 * N * th * tw * cin * cc * k * k
kw = dict(stride=1, pad=1, aux_per_image=True, ps=(s, cc, g0, Hf, Wf), precision="bf16x3", n_alg=k * k * cc, out_split="only", out_il=ops.split_il(cc))
kw.update(dict(epi="ps_add2", aux1=off, aux2=res) if a.v2p else dict(act="relu", epi="affine", aux1=off, aux2=sc))


def timed(fn):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / a.reps


print(f"fold convolution {'Vec2Patch' if a.v2p else 'FFN'}: {N} frames x {th}x{tw} tokens, {cin} -> {k * k}x{cc}, Cout {cout} (g0 {g0}); credited {flops / 1e9:.1f} GFLOP per launch")
ref = None
for tile in ("128x128it", "256x128it", "128x128x8t", "128x128t", "128x64t", "128x64x8t", "256x256it"):
    row = []
    for order in (0, 1):
        for skip in (g0, 0):
            try:
                ms = timed(lambda: ops.conv2d(x4, pc, tile=tile, ky_skip_n0=skip, tile_order=order, **kw))
            except RuntimeError as e:
                row.append(f"order {order} skip {int(bool(skip))}: n/a")
                continue
            out = ops.conv2d(x4, pc, tile=tile, ky_skip_n0=skip, tile_order=order, **kw)
            ref = out if ref is None else ref
            same = torch.equal(out.data, ref.data)
            row.append(f"order {order} skip {int(bool(skip))}: {ms:6.3f} ms {flops / ms / 1e9:6.1f} TF{'' if same else ' DIFFERS'}")
    print(f"{tile:11s} " + " | ".join(row))
# the formulation it replaces
w1p = w.view(cc, k * k, cin).permute(1, 0, 2).reshape(cc * k * k, cin)
b1p = b.view(cc, k * k).permute(1, 0).reshape(-1)
pl = ops.PackedConv(w1p, b1p)
ms_l = timed(lambda: ops.linear(xs, pl))
Y = ops.linear(xs, pl)
ms_f = timed(lambda: ops.fold(Y, N, th, tw, cc, k, s, p, Hf, Wf, normalize=not a.v2p, res=res, relu=not a.v2p, out_split=True))
print(f"Linear (autotuned tile) {ms_l:6.3f} ms {flops / ms_l / 1e9:6.1f} TF + fgt_fold {ms_f:6.3f} ms = {ms_l + ms_f:6.3f} ms")
