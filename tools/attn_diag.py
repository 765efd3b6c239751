#!/usr/bin/env python
"""Diagnostic: where does the bf16x3 temporal attention differ from the fp32 kernel on long zones?  (run with FGT_ATTN_NW8=0 / 1)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (t, nh, nw, scale) in ((13, 20, 36, 1.5), (5, 22, 36, 1.5), (13, 20, 36, 0.05), (12, 20, 36, 1.5), (11, 20, 36, 1.5)):
    g = torch.Generator().manual_seed(100 + t)
    qkv = torch.randn(t * nh * nw, 1536, generator=g)
    qkv[:, :1024] *= scale
    x = qkv.to(dev)
    a = ops.attention_temporal(x, 1, t, nh, nw, 4, 2, 512, precision="fp32")
    b = ops.attention_temporal(x, 1, t, nh, nw, 4, 2, 512, precision="bf16x3")
    e = (a - b).abs()                                   # [t*nh*nw, 512]
    zh, zw = nh // 2, nw // 2
    L = t * zh * zw
    print(f"t={t} {nh}x{nw} scale={scale} L={L} NW8={os.environ.get('FGT_ATTN_NW8', '1')}: max err {e.max().item():.3e} (ref max {a.abs().max().item():.2f}), frac rows > 1e-3: {(e.max(1)[0] > 1e-3).float().mean().item():.3f}")
    ez = e.view(t, 2, zh, 2, zw, 4, 128).permute(1, 3, 5, 0, 2, 4, 6).reshape(4, 4, L, 128)        # [zone, head, local token, d]
    per_tok = ez.amax(-1)                                # [zone, head, L]
    print("  per (zone, head) max:", [[round(v, 3) for v in r] for r in per_tok.amax(-1).tolist()])
    blk = per_tok.amax((0, 1))                           # [L]
    per32 = [round(blk[i:i + 32].max().item(), 3) for i in range(0, L, 32)]
    print("  per 32-query group (first 24):", per32[:24])
    print("  per 32-query group (last 12):", per32[-12:])
    print("  per d-block of 32:", [round(ez[..., i:i + 32].max().item(), 3) for i in range(0, 128, 32)])
