#!/usr/bin/env python
"""Run-to-run determinism of the flow stages under GPU sharing (start several side by side): RAFT over 9 frames at 864x480 (16 pairs, 20 iterations)
and LAFC over the clip's first 24 flows, every pass compared bit for bit with the first.    python tools/determinism_flow_check.py [--passes 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_stages  # noqa: E402
from fgt_amd import flow_pipeline, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=10)
a = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = "bf16x3"
lafc, lsd, raft, rsd = bench_stages._models(dev)
inp = bench_stages.stage_inputs(26, 240, 432)
v2 = torch.nn.functional.interpolate(inp["video"][:9].to(dev), size=(480, 864), mode="bilinear", align_corners=False)
ffl = inp["flow_f"][:24].to(dev).permute(1, 0, 2, 3)[None].contiguous()
mk = inp["hole"][:24].float().to(dev)[None, None]
dif = flow_pipeline.diffusion(ffl, mk)


def once():
    f, b = flow_pipeline.compute_flows(raft, v2, iters=20)
    c = flow_pipeline.complete_flows(lafc, ffl, mk, dif)
    torch.cuda.synchronize()
    return f.clone(), b.clone(), c.clone()


once()
ref = once()
bad = [0, 0, 0]
for i in range(a.passes):
    cur = once()
    for j in range(3):
        if not torch.equal(cur[j], ref[j]):
            bad[j] += 1
            print(f"pass {i}: output {j} differs, max |diff| {float((cur[j] - ref[j]).abs().max()):.3e}")
print(f"pid {os.getpid()}: {a.passes} passes; differing from the first: RAFT forward {bad[0]}, RAFT backward {bad[1]}, LAFC {bad[2]}")
