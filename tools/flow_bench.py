#!/usr/bin/env python
"""BASELINE config #4 / #5 timings on the GPU box: LAFC call and RAFT pair at 432x240 (and the tool-faithful 864x480
RAFT input), plus one FGT window of the 864x480x160 clip (t = 26, 2880 tokens/frame).  Prints one JSON line per case and
writes gpurun_out/flow_bench.json.  CPU oracle timings are taken on small bounded samples."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402
from fgt_amd.synth import synth_clip, synth_state_dict  # noqa: E402


def timed(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--cpu", action="store_true", help="also time the CPU oracle on the same inputs")
    a = ap.parse_args()
    ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = a.precision
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    out = []
    # ---- LAFC: 3 flows at 240x432 (130.2 GFLOP per call)
    from fgt_amd import lafc_model
    m = lafc_model.Model(dict(lafc_model.DEFAULT_CONFIG)).eval()
    sd = synth_state_dict(m.state_dict(), seed=0, mode="kaiming")
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    g = torch.Generator().manual_seed(0)
    fl = torch.randn(1, 2, 3, 240, 432, generator=g)
    ms = (torch.rand(1, 1, 3, 240, 432, generator=g) > 0.8).float()
    dfl, dms = fl.to(dev), ms.to(dev)
    t = timed(lambda: m(dfl, dms), 10)
    rec = {"case": "LAFC forward [1,2,3,240,432]", "ms": round(t * 1e3, 3), "tflops_algorithmic": round(130.2e9 / t / 1e12, 2), "precision": a.precision}
    if a.cpu:
        from oracle import lafc_oracle as LO
        t0 = time.perf_counter(); ref = LO.lafc_forward(sd, lafc_model.DEFAULT_CONFIG, fl, ms); rec["cpu_oracle_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        rec["max_abs_vs_oracle"] = float((m(dfl, dms)[0].cpu() - ref[0]).abs().max())
    out.append(rec)
    # ---- RAFT pairs, 20 iterations
    import argparse as ap2
    from fgt_amd import raft_model
    r = raft_model.RAFT(ap2.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
    rsd = synth_state_dict(r.state_dict(), seed=0, mode="kaiming")
    r.load_state_dict(rsd, strict=True)
    r = r.to(dev)
    for (H, W, gf) in ((240, 432, 245.7), (480, 864, 998.9)):
        base = torch.nn.functional.interpolate(torch.rand(1, 3, H // 8 + 2, W // 8 + 2, generator=g), size=(H + 8, W + 8), mode="bilinear") * 255
        i1, i2 = base[:, :, 4:4 + H, 4:4 + W].contiguous().to(dev), base[:, :, 3:3 + H, 6:6 + W].contiguous().to(dev)
        t = timed(lambda: r(i1, i2, iters=20, test_mode=True), 5)
        out.append({"case": f"RAFT pair {W}x{H}, 20 iters", "ms": round(t * 1e3, 3), "tflops_algorithmic": round(gf * 1e9 / t / 1e12, 2), "precision": a.precision})
    # ---- clip-level RAFT: per-frame encoder cache + batched pairs (fgt_amd/flow_pipeline.py), 16 frames = 30 pairs
    from fgt_amd import flow_pipeline
    for (H, W, gf) in ((240, 432, 245.7), (480, 864, 998.9)):
        frames = torch.nn.functional.interpolate(torch.rand(16, 3, H // 8, W // 8, generator=g), size=(H, W), mode="bilinear").to(dev) * 255
        for bsz in (1, 8):
            t = timed(lambda: flow_pipeline.compute_flows(r, frames, iters=20, batch=bsz), 2)
            out.append({"case": f"RAFT clip pipeline {W}x{H}, 16 frames (30 pairs), pair batch {bsz}", "ms_per_pair": round(t * 1e3 / 30, 3),
                        "tflops_algorithmic_vs_reference_count": round(30 * gf * 1e9 / t / 1e12, 2), "precision": a.precision})
    # ---- clip-level LAFC: batched pivots (fgt_amd/flow_pipeline.complete_flows), 32 flows at 240x432
    flows32 = torch.randn(1, 2, 32, 240, 432, generator=g).to(dev)
    masks32 = (torch.rand(1, 1, 32, 240, 432, generator=g) > 0.8).float().to(dev)
    for bsz in (1, 8):
        t = timed(lambda: flow_pipeline.complete_flows(m, flows32, masks32, flows32 * (1 - masks32), batch=bsz), 2)
        out.append({"case": f"LAFC clip pipeline 432x240, 32 flows, pivot batch {bsz}", "ms_per_flow": round(t * 1e3 / 32, 3),
                    "tflops_algorithmic": round(32 * 130.2e9 / t / 1e12, 2), "precision": a.precision})
    # ---- one FGT window of BASELINE config #5: 864x480, t = 26
    from fgt_amd.fgt_model import DEFAULT_CONFIG, Model
    f = Model(dict(DEFAULT_CONFIG)).eval()
    f.load_state_dict(synth_state_dict(f.state_dict(), seed=0), strict=True)
    f = f.to(dev)
    fr, fw, mk = synth_clip(26, 480, 864, device=dev)
    mf = (fr * 2 - 1) * (1 - mk)
    t = timed(lambda: f(mf, fw, mk), 3)
    out.append({"case": "FGT window 864x480, t=26 (config #5 unit)", "ms": round(t * 1e3, 2), "frames_per_s_this_window": round(26 / t, 1),
                "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "precision": a.precision})
    # ---- diffusion fill of a clip's flows (tool/video_inpainting.py:42-51): 79 flows x 2 channels at 432x240, object-like holes
    import numpy as np
    from fgt_amd import flow_pipeline as FP
    tt, H, W = 79, 240, 432
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    mk = torch.stack([(((yy - 120 - 0.3 * i) / 70.0) ** 2 + ((xx - 150 - 1.5 * i) / 80.0) ** 2 <= 1.0).float() for i in range(tt)])[None, None]
    fl2 = torch.cumsum(torch.randn(1, 2, tt, H, W, generator=g), -1) * 0.3
    dfl2, dmk = fl2.to(dev), mk.to(dev)
    for iters in (300, 1000):
        t = timed(lambda: FP.diffusion(dfl2, dmk, iters=iters), 2)
        rec = {"case": f"diffusion fill, {tt} flows x 2 channels at {W}x{H}, ~{int(mk[0, 0, 0].sum())} px holes, {iters} CG iterations",
               "ms_per_direction": round(t * 1e3, 2), "ms_per_flow": round(t * 1e3 / tt, 3)}
        if a.cpu:
            from oracle import fill_oracle as FO
            got = FP.diffusion(dfl2, dmk, iters=iters)[0].permute(1, 2, 3, 0).cpu().numpy()        # [t,H,W,2]
            t0 = time.perf_counter()
            ref = np.stack([FO.regionfill(fl2[0, c, i].numpy(), mk[0, 0, i].numpy()) for i in range(3) for c in range(2)])
            rec["cpu_oracle_ms_per_flow"] = round((time.perf_counter() - t0) * 1e3 / 3, 1)
            gg = np.stack([got[i, :, :, c] for i in range(3) for c in range(2)])
            rec["max_abs_vs_oracle_first3"] = float(np.abs(gg - ref).max())
            rec["value_range"] = float(np.abs(ref).max())
        out.append(rec)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/flow_bench.json", "w"), indent=1)
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
