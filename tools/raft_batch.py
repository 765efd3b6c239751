#!/usr/bin/env python
"""RAFT clip pipeline (flow_pipeline.compute_flows) vs pair-batch size: ms per pair at 432x240 and 864x480, 20 iterations."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import flow_pipeline, ops, raft_model  # noqa: E402
from fgt_amd.synth import synth_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=33)
ap.add_argument("--batches", default="8,16,32")
a = ap.parse_args()
ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = "bf16x3"
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
r = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
r.load_state_dict(synth_state_dict(r.state_dict(), seed=0, mode="kaiming"), strict=True)
r = r.to(dev)
g = torch.Generator().manual_seed(0)
for (H, W) in ((240, 432), (480, 864)):
    frames = torch.nn.functional.interpolate(torch.rand(a.frames, 3, H // 8, W // 8, generator=g), size=(H, W), mode="bilinear").to(dev) * 255
    ref = None
    for b in [int(x) for x in a.batches.split(",")]:
        flow_pipeline.compute_flows(r, frames, iters=20, batch=b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fw, bw = flow_pipeline.compute_flows(r, frames, iters=20, batch=b)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = "" if ref is None else f" bit-equal to batch {ref[0]}: {torch.equal(fw, ref[1]) and torch.equal(bw, ref[2])}"
        if ref is None:
            ref = (b, fw, bw)
        print(f"RAFT {W}x{H} {a.frames} frames ({2 * (a.frames - 1)} pairs), pair batch {b}: {dt * 1e3 / (2 * (a.frames - 1)):.3f} ms per pair, "
              f"peak mem {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GB{same}", flush=True)
