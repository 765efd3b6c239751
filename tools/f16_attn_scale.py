"""Spatial / temporal fp16 attention at the bench's batched-window sizes (bt = 136 frames of 20x36 tokens, compact maps, global tokens
in the tail of the k / v buffers) vs the bf16x3 split kernel on the same fp16 values.  python tools/f16_attn_scale.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
heads, ws, gd, c = 4, 8, 4, 512
for bt in (13, 64, 136, 144):
    h, w, nh, nw = 20, 36, 24, 40
    ng = (nh // gd) * (nw // gd)
    R = bt * h * w
    q32 = torch.randn(R + 1, c, device=dev)
    k32 = torch.randn(R + 1 + bt * ng, c, device=dev)
    v32 = torch.randn(R + 1 + bt * ng, c, device=dev)
    outs = {}
    for fmt in (True, False):
        # the bf16-pair run gets the fp16-rounded values, so both kernels see the same operands
        mk = lambda x: ops.split(x.half().float(), h=fmt)
        q, k, v = mk(q32), mk(k32), mk(v32)
        for rep in range(3):
            o = ops.attention_spatial(q, k[:R + 1], v[:R + 1], k[R + 1:], v[R + 1:], bt, h, w, nh, nw, heads, ws, ng, pad_row=R)
        torch.cuda.synchronize()
        outs[fmt] = o
    d = (outs[True] - outs[False]).abs().max().item()
    print(f"spatial bt={bt}: f16 vs bf16x3 on the same fp16 values: max diff {d:.3e} (out max {outs[False].abs().max().item():.3f})", flush=True)
for b, t in ((8, 17), (7, 18)):
    nh, nw = 20, 36
    qkv32 = torch.randn(b * t * nh * nw, 3 * c, device=dev)
    outs = {}
    for fmt in (True, False):
        sp = ops.split(qkv32.half().float(), h=fmt)
        for rep in range(2):
            o = ops.attention_temporal(sp, b, t, nh, nw, heads, 2, c, tq=11)
        torch.cuda.synchronize()
        outs[fmt] = o
    d = (outs[True] - outs[False]).abs().max().item()
    print(f"temporal b={b} t={t} tq=11: max diff {d:.3e}", flush=True)
print("OK")
