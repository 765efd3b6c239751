#!/usr/bin/env python
"""RAFT over the bench clip (80 frames -> 158 pairs, 20 iterations) through flow_pipeline.compute_flows at 864x480: ms per pair and a checksum, for
same-box A/Bs of the environment switches (FGT_RAFT_BATCH_STREAMS, FGT_RAFT_STREAMS, FGT_RAFT_ZR, FGT_RAFT_BCORR).   python tools/raft_clip.py [--reps 3]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_stages  # noqa: E402
from fgt_amd import flow_pipeline, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--width", type=int, default=864)
ap.add_argument("--batch", type=int, default=None)
a = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = "bf16x3"
inp = bench_stages.stage_inputs(80, a.height // 2, a.width // 2)
v = torch.nn.functional.interpolate(inp["video"], size=(a.height, a.width), mode="bilinear", align_corners=False).to(dev)
raft = bench_stages._models(dev)[2]
fw, bw = flow_pipeline.compute_flows(raft, v, iters=20, batch=a.batch)
torch.cuda.synchronize()
ts = []
for _ in range(a.reps):
    t0 = time.perf_counter()
    fw, bw = flow_pipeline.compute_flows(raft, v, iters=20, batch=a.batch)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ms = min(ts) * 1e3
env = {k: os.environ.get(k) for k in ("FGT_RAFT_BATCH_STREAMS", "FGT_RAFT_STREAMS", "FGT_RAFT_ZR", "FGT_RAFT_BCORR") if os.environ.get(k) is not None}
print(f"RAFT {a.width}x{a.height} 158 pairs x 20 iterations: {ms:.1f} ms per clip, {ms / 158:.3f} ms per pair (best of {a.reps}), "
      f"checksum {float(fw.double().abs().mean()):.6f} / {float(bw.double().abs().mean()):.6f}  {env} batch={a.batch}")
