#!/usr/bin/env python
"""One SWMHSA module at BASELINE config C2 (t = 10, 20x36 tokens) as a captured hipGraph, one stream vs three: wall time per replay, and — under
`rocprofv3 --kernel-trace` — the kernel timeline (tools/c2_timeline.py reads the CSV: which launches overlap).
  python tools/c2_module.py [--prec bf16x3|fp32] [--t 10] [--streams 0|1] [--reps 20]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import fgt_model, ops  # noqa: E402
from fgt_amd.fgt_model import DEFAULT_CONFIG, Model  # noqa: E402
from fgt_amd.graph import GraphedCall  # noqa: E402
from fgt_amd.synth import synth_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--prec", default="bf16x3")
ap.add_argument("--t", type=int, default=10)
ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--eager", action="store_true")
a = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = a.prec
fgt_model.SPATIAL_STREAM_ROWS = 1 << 30 if a.streams else 0
fgt_model.SPATIAL_STREAMS_EAGER = bool(a.streams)
m = Model(dict(DEFAULT_CONFIG)).eval()
m.load_state_dict(synth_state_dict(m.state_dict(), seed=0), strict=True)
net = m.to(dev).net
P = net.packed()
th, tw = 20, 36
g = torch.Generator().manual_seed(1234)
x, f = torch.randn(a.t * th * tw, 512, generator=g).to(dev), torch.randn(a.t * th * tw, 256, generator=g).to(dev)
fn = lambda u, v: net._spatial_attention(u, v, P["s0"], a.t, th, tw)
if a.eager:
    run = lambda: fn(x, f)
else:
    gc = GraphedCall(fn, [x, f])
    run = gc.graph.replay
for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
print(f"C2 SWMHSA {a.prec} t={a.t} streams={'3' if a.streams else '1'} {'eager' if a.eager else 'graph'}: {dt * 1e3:.4f} ms per call")
