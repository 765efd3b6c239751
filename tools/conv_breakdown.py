#!/usr/bin/env python
"""Per-layer-shape time of every fgt_conv2d launch of one clip pass (bench configuration), with algorithmic TFLOP/s and the
unique-bytes floor (input + weights + output bytes once) per shape.

    python tools/conv_breakdown.py [--precision bf16x3] [--frames 80]
"""
import argparse
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402
from fgt_amd.fgt_model import DEFAULT_CONFIG, Model  # noqa: E402
from fgt_amd.scheduler import ClipRunner  # noqa: E402
from fgt_amd.synth import synth_clip, synth_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--frames", type=int, default=80)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=432)
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = a.precision
    m = Model(dict(DEFAULT_CONFIG)).eval()
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0), strict=True)
    m = m.to(dev)
    fr, fl, ms = synth_clip(a.frames, a.height, a.width, seed=1234, device=dev)
    r = ClipRunner(m, fr, fl, ms)
    r.run(); r.run()
    torch.cuda.synchronize()
    recs = []
    real = ops.conv2d

    def wrapped(x, pc, *args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = real(x, pc, *args, **kw)
        e1.record()
        xs = x.shape
        x1 = kw.get("x1")
        o = out[0] if isinstance(out, tuple) else out
        key = (tuple(xs), 0 if x1 is None else x1.shape[-1], pc.Cout, pc.groups, f"{pc.kh}x{pc.kw}", kw.get("stride", 1), ("f16" if x.h else "split") if isinstance(x, ops.Split) else "fp32",
               {None: "f32", "only": "split", "both": "f32+split"}[kw.get("out_split")], (kw.get("epi") or "-") + ("/ps" if kw.get("ps") else ""))
        osh = tuple(o.shape)
        M = osh[0] * osh[1] * osh[2] if len(osh) == 4 else osh[0]
        cin = (xs[-1] + (0 if x1 is None else x1.shape[-1]))
        flops = 2.0 * M * (pc.Cout // pc.groups) * pc.k_alg * pc.groups
        if kw.get("ps"):                      # fold as a convolution: credited with the Linear's work (rows x cin x k*k*cc), written map = Hf x Wf x cc
            flops = 2.0 * xs[0] * xs[1] * xs[2] * kw["n_alg"] * pc.k_alg
        h_in = isinstance(x, ops.Split) and x.h                        # fp16 tensors: 2 B per value
        osp = kw.get("out_split")
        h_out = bool(osp) and a.precision == "f16"
        in_b = (2.0 if h_in else 4.0) * xs[0] * xs[1] * xs[2] * cin
        out_b = M * pc.Cout * ((4.0 if osp != "only" else 0.0) + ((2.0 if h_out else 4.0) if osp else 0.0)) + (4.0 * M * pc.Cout if kw.get("epi") else 0)
        w_b = (2.0 if h_in else 4.0) * pc.Cout * pc.K
        recs.append((key, e0, e1, flops, in_b + out_b + w_b))
        return out

    ops.conv2d = wrapped
    import fgt_amd.fgt_model as fm
    fm.ops.conv2d = wrapped
    r.run()
    torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for key, e0, e1, fl_, by in recs:
        g = agg[key]
        g[0] += 1; g[1] += e0.elapsed_time(e1); g[2] += fl_; g[3] += by
    tot = sum(v[1] for v in agg.values())
    print(f"{len(recs)} conv launches, {tot:.2f} ms (event-bracketed, includes launch gaps)")
    print(f"{'input':28s} {'C1':>4s} {'Cout':>5s} {'g':>2s} {'k':>4s} {'s':>2s} {'in':>6s} {'out':>9s} {'epi':>10s} {'calls':>5s} {'ms':>8s} {'%':>5s} {'TF alg':>7s} {'GB/s floor':>10s}")
    for key, (n, ms_, fl_, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{str(key[0]):28s} {key[1]:4d} {key[2]:5d} {key[3]:2d} {key[4]:>4s} {str(key[5]):>2s} {key[6]:>6s} {key[7]:>9s} {key[8]:>10s} {n:5d} {ms_:8.3f} {100 * ms_ / tot:5.1f} {fl_ / ms_ / 1e9:7.1f} {by / ms_ / 1e6:10.0f}")


if __name__ == "__main__":
    main()
