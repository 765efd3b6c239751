#!/usr/bin/env python
"""Where a LAFC completion batch spends its time: every fgt_amd.ops call of one forward (8 pivots x 3 flows at 432x240 by default)
bracketed with HIP events, aggregated per (op, shape).    python tools/lafc_breakdown.py [--height 240 --width 432 --pivots 8]"""
import argparse
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import lafc_model, ops  # noqa: E402
from fgt_amd.synth import synth_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=240)
ap.add_argument("--width", type=int, default=432)
ap.add_argument("--pivots", type=int, default=8)
a = ap.parse_args()
ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = "bf16x3"
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
m = lafc_model.Model(dict(lafc_model.DEFAULT_CONFIG)).eval()
m.load_state_dict(synth_state_dict(m.state_dict(), seed=0, mode="kaiming"), strict=True)
m = m.to(dev)
g = torch.Generator().manual_seed(0)
B, T = a.pivots, 3
flows = torch.randn(B, 2, T, a.height, a.width, generator=g).to(dev)
masks = (torch.rand(B, 1, T, a.height // 8, a.width // 8, generator=g) > 0.7).float().repeat_interleave(8, 3).repeat_interleave(8, 4).to(dev)
for _ in range(2):
    m(flows * (1 - masks), masks)
torch.cuda.synchronize()
recs = []
NAMES = ["conv2d", "axpby", "nchw_to_nhwc", "nhwc_to_nchw", "split"]
real = {k: getattr(ops, k) for k in NAMES}


def wrap(name):
    fn = real[name]

    def w(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kw)
        e1.record()
        x = args[0]
        key, flops = name, 0.0
        if name == "conv2d":
            pc = args[1]
            x1 = kw.get("x1")
            o = out[0] if isinstance(out, tuple) else out
            M = 1
            for d_ in tuple(o.shape)[:-1]:
                M *= d_
            flops = 2.0 * M * (pc.Cout // pc.groups) * pc.k_alg * pc.groups
            key = (f"{name} {tuple(x.shape)} +{0 if x1 is None else x1.shape[-1]} -> {pc.Cout} k{pc.kh}x{pc.kw} s{kw.get('stride', 1)} d{kw.get('dil', 1)} "
                   f"{'up ' if kw.get('upsample') else ''}{type(x).__name__[0]}>{(kw.get('out_split') or 'f32')[0]} {kw.get('act') or '-'} {kw.get('epi') or '-'}")
        elif hasattr(x, "shape"):
            key = f"{name} {tuple(x.shape)}"
        recs.append((key, e0, e1, flops))
        return out
    return w


for k in NAMES:
    setattr(ops, k, wrap(k))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
m(flows * (1 - masks), masks)
e1.record()
torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0, 0.0])
for key, a0, a1, fl in recs:
    v = agg[key]
    v[0] += 1; v[1] += a0.elapsed_time(a1); v[2] += fl
tot = sum(v[1] for v in agg.values())
print(f"LAFC forward {B} pivots x {T} flows {a.width}x{a.height}: {e0.elapsed_time(e1):.2f} ms wall, {tot:.2f} ms in {len(recs)} bracketed ops "
      f"({e0.elapsed_time(e1) / B:.3f} ms per completed flow)")
for key, (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms:9.3f} ms {100 * ms / tot:5.1f}% {cnt:5d}x {fl / ms / 1e9 if ms else 0:7.1f} TF  {key}")
