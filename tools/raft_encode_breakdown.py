#!/usr/bin/env python
"""Where RAFT's per-frame encoders (fnet: InstanceNorm, cnet: BatchNorm folded) spend their time: every fgt_amd.ops call of encode_features /
encode_context on 16 frames of 864x480 bracketed with HIP events, aggregated per (op, shape).   python tools/raft_encode_breakdown.py [--frames 16]"""
import argparse
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops, raft_model  # noqa: E402
from fgt_amd.synth import synth_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--width", type=int, default=864)
a = ap.parse_args()
ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = "bf16x3"
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
r = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
r.load_state_dict(synth_state_dict(r.state_dict(), seed=0, mode="kaiming"), strict=True)
r = r.to(dev)
frames = torch.rand(a.frames, 3, a.height, a.width, generator=torch.Generator().manual_seed(0)).to(dev) * 255
packed = r.pack_images(frames)
for _ in range(2):
    r.encode_features(packed); r.encode_context(packed)
torch.cuda.synchronize()
recs = []
NAMES = ["conv2d", "instnorm"]
real = {k: getattr(ops, k) for k in NAMES}


def wrap(name):
    fn = real[name]

    def w(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kw)
        e1.record()
        x = args[0]
        key, fl = f"{name} {tuple(x.shape)}", 0.0
        if name == "conv2d":
            pc = args[1]
            o = out[0] if isinstance(out, tuple) else out
            M = 1
            for d_ in tuple(o.shape)[:-1]:
                M *= d_
            fl = 2.0 * M * pc.Cout * pc.k_alg
            key += f" -> {pc.Cout} k{pc.kh} s{kw.get('stride', 1)}"
        recs.append((key, e0, e1, fl))
        return out
    return w


for k in NAMES:
    setattr(ops, k, wrap(k))
for which, fn in (("fnet (InstanceNorm)", r.encode_features), ("cnet (BatchNorm folded)", r.encode_context)):
    recs.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(packed); e1.record(); torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for key, a0, a1, fl in recs:
        v = agg[key]; v[0] += 1; v[1] += a0.elapsed_time(a1); v[2] += fl
    tot = sum(v[1] for v in agg.values())
    by = defaultdict(float)
    for key, v in agg.items():
        by[key.split()[0]] += v[1]
    print(f"{which}: {a.frames} frames {a.width}x{a.height}: {e0.elapsed_time(e1):.2f} ms wall ({e0.elapsed_time(e1) / a.frames:.3f} ms per frame); " + ", ".join(f"{k} {v:.2f} ms" for k, v in by.items()))
    for key, (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"{ms:9.3f} ms {100 * ms / tot:5.1f}% {cnt:4d}x {fl / ms / 1e9 if ms else 0:7.1f} TF  {key}")
