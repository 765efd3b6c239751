#!/usr/bin/env python
"""The flow stages' bandwidth-bound kernels at the bench's shapes, as a target for `rocprofv3 --pmc` (tools/gpu_check.sh pmc stage):
  warp         fbConsistencyCheck's two image_warp calls over the clip's 79 flow pairs + one 3-channel frame warp (bench_stages.py, 432x240)
  corr_lookup  RAFT's 4-level 9x9 lookup inside a real refinement loop: 8 pairs at 864x480, 20 iterations
Prints (and writes to argv[1], default gpurun_out/hbm_micro_alg.json) the ALGORITHMIC bytes per launch the library's own accounting credits
(fgt_prof_*), which tools/pmc_traffic.py --alg stores beside the counter bytes.

    python tools/hbm_micro.py [out.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_stages  # noqa: E402
from fgt_amd import flow_pipeline, ops  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join("gpurun_out", "hbm_micro_alg.json")
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = "bf16x3"
    N, H, W = 80, 240, 432
    g = torch.Generator().manual_seed(7)
    inp = bench_stages.stage_inputs(N, H, W)                   # the bench's own clip: smooth flows of a few pixels (white-noise flows would scatter the taps)
    flf = inp["flow_f"].permute(0, 2, 3, 1).contiguous().to(dev)
    flb = inp["flow_b"].permute(0, 2, 3, 1).contiguous().to(dev)
    img = inp["img"][: N - 1].contiguous().to(dev)
    raft = bench_stages._models(dev)[2]
    video = (torch.rand(5, 3, 2 * H, 2 * W, generator=g) * 255).to(dev)
    flow_pipeline.compute_flows(raft, video, iters=2)          # packing + tile tuning outside the measured part
    torch.cuda.synchronize()
    ops.prof_collect("all")
    ops.prof_enable(True)
    for _ in range(3):
        ops.warp(flf, flb)
        ops.warp(flb, flf)
        ops.warp(img, flf)
    flow_pipeline.compute_flows(raft, video, iters=20)
    torch.cuda.synchronize()
    ops.prof_enable(False)
    alg = {}
    for kind in ("warp", "corr_lookup"):
        ms, fl, n, by = ops.prof_collect(kind)
        alg[kind] = {"algorithmic_bytes_per_launch": round(by / max(n, 1)), "event_ms_per_launch": round(ms / max(n, 1), 4), "launches_timed": n,
                     "algorithmic_GBps": round(by / (ms * 1e-3) / 1e9, 1) if ms > 0 else None}
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    json.dump(alg, open(out_path, "w"), indent=1)
    print(json.dumps(alg))


if __name__ == "__main__":
    main()
