#!/usr/bin/env python
"""Where a RAFT refinement batch spends its time: every fgt_amd.ops call of one `iterate` (32 pairs, 20 iterations, 864x480 by
default) bracketed with HIP events, aggregated per (op, shape).    python tools/raft_breakdown.py [--height 480 --width 864 --pairs 32]"""
import argparse
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops, raft_model  # noqa: E402
from fgt_amd.synth import synth_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--width", type=int, default=864)
ap.add_argument("--pairs", type=int, default=32)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = "bf16x3"
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
r = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
r.load_state_dict(synth_state_dict(r.state_dict(), seed=0, mode="kaiming"), strict=True)
r = r.to(dev)
g = torch.Generator().manual_seed(0)
n = a.pairs + 1
frames = torch.nn.functional.interpolate(torch.rand(n, 3, a.height // 8, a.width // 8, generator=g), size=(a.height, a.width), mode="bilinear").to(dev) * 255
packed = r.pack_images(frames)
fmap, cmap = r.encode_features(packed), r.encode_context(packed)
i1, i2 = torch.arange(0, n - 1, device=dev), torch.arange(1, n, device=dev)
for _ in range(2):
    r.iterate(fmap[i1], fmap[i2], cmap[i1], iters=a.iters, test_mode=True)
torch.cuda.synchronize()
recs = []
NAMES = ["conv2d", "linear", "corr_lookup", "batched_gemm_nt", "split", "axpby", "avgpool2", "convex_upsample", "nhwc_to_nchw", "instnorm"]
real = {k: getattr(ops, k) for k in NAMES}


def wrap(name):
    fn = real[name]

    def w(*args, **kw):
        if name == "linear":                      # linear calls conv2d: time it as one op
            saved = ops.conv2d
            ops.conv2d = real["conv2d"]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kw)
        e1.record()
        if name == "linear":
            ops.conv2d = saved
        x = args[0]
        key, flops = name, 0.0
        if name in ("conv2d", "linear"):
            pc = args[1]
            x1 = kw.get("x1")
            o = out[0] if isinstance(out, tuple) else out
            M = 1
            for d_ in tuple(o.shape)[:-1]:
                M *= d_
            flops = 2.0 * M * (pc.Cout // pc.groups) * pc.k_alg * pc.groups
            key = f"{name} {tuple(x.shape)} +{0 if x1 is None else x1.shape[-1]} -> {pc.Cout} k{pc.kh}x{pc.kw} {kw.get('act') or '-'} {kw.get('epi') or '-'}"
        elif name == "batched_gemm_nt":
            G, M, K = x.shape
            flops = 2.0 * G * M * args[1].shape[1] * K
            key = f"batched_gemm_nt {tuple(x.shape)} x {tuple(args[1].shape)}"
        elif hasattr(x, "shape"):
            key = f"{name} {tuple(x.shape)}"
        recs.append((key, e0, e1, flops))
        return out
    return w


for k in NAMES:
    setattr(ops, k, wrap(k))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
r.iterate(fmap[i1], fmap[i2], cmap[i1], iters=a.iters, test_mode=True)
e1.record()
torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0, 0.0])
for key, a0, a1, fl in recs:
    v = agg[key]
    v[0] += 1; v[1] += a0.elapsed_time(a1); v[2] += fl
tot = sum(v[1] for v in agg.values())
print(f"RAFT iterate {a.pairs} pairs {a.width}x{a.height}, {a.iters} iterations: {e0.elapsed_time(e1):.1f} ms wall, {tot:.1f} ms in {len(recs)} bracketed ops "
      f"({e0.elapsed_time(e1) / a.pairs:.3f} ms per pair)")
for key, (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms:9.3f} ms {100 * ms / tot:5.1f}% {cnt:5d}x {fl / ms / 1e9 if ms else 0:7.1f} TF  {key}")
