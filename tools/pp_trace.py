#!/usr/bin/env python
"""Per-wavefront timeline of the interleaved-request tap kernel (csrc/conv_taps_il.hip; diagnostic build with -DFGT_PP_TRACE).

    python tools/pp_trace.py --build            # here (no GPU needed): lib/libfgt_hip_pptrace.so
    python tools/pp_trace.py [--layer e20enc10] [--tile 256x128it]      # on the MI355X

Stamps per step (s_memtime cycles): 0 step top | 5 first MFMA about to issue | 6 last MFMA issued | 3 requests waited for (vmcnt) | 4 behind the barrier."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "fgt_amd", "lib", "libfgt_hip_pptrace.so")
STEPS = 24
LAYERS = {"e20enc10": (20, 60, 108, 256, 384, 512, 2, 3, 1, 1), "e20enc8": (20, 60, 108, 256, 0, 384, 1, 3, 1, 1), "raftgru": (32, 60, 108, 128, 256, 128, 1, (1, 5), 1, (0, 2))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--layer", default="e20enc10")
    ap.add_argument("--tile", default="256x128it")
    ap.add_argument("--il", action="store_true", help="interleaved split inputs (the wide LDS image)")
    a = ap.parse_args()
    if a.build:
        from fgt_amd import build
        print(build.build(variant="pptrace", extra_flags=["-DFGT_PP_TRACE"]))
        return
    os.environ["FGT_HIP_LIB"] = LIB
    import numpy as np
    import torch
    from fgt_amd import _lib, ops
    h = _lib.lib()
    dev = torch.device("cuda:0")
    N, H, W, C0, C1, Cout, g, k, s, p = LAYERS[a.layer]
    kh, kw = (k, k) if isinstance(k, int) else k
    x = torch.randn(N, H, W, C0, device=dev)
    x1 = torch.randn(N, H, W, C1, device=dev) if C1 else None
    w = torch.randn(Cout, (C0 + C1) // g, kh, kw, device=dev) * 0.02
    pc = ops.PackedConv(w, torch.zeros(Cout, device=dev), groups=g)
    xs, x1s = ops.split(x, interleave=a.il), (ops.split(x1, interleave=a.il) if C1 else None)        # --il: the wide LDS image (8-row x 128-byte pieces)
    for _ in range(5):
        ops.conv2d(xs, pc, x1=x1s, stride=s, pad=p, act="lrelu", tile=a.tile, precision="bf16x3")
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (8 * STEPS * 10))()
    assert h.fgt_debug_pp_trace(buf, 8 * STEPS * 10) == 0
    t = np.array(buf, dtype=np.int64).reshape(8, STEPS, 10)
    t0 = t[:, 2:, :]                       # skip the first two steps (cold)
    base = t0[:, :, 0].min()
    print(f"{a.layer} {a.tile}: per wavefront, median cycles over steps 2..{STEPS - 1}")
    print("wave | top -> first MFMA | MFMAs (+ interleaved requests) | tail + vmcnt | barrier | step period")
    for wv in range(8):
        if t0[wv].max() == 0:
            continue                                             # (4-wavefront tiles)
        med = lambda a, b: np.median(t0[wv, :, b] - t0[wv, :, a])
        period = np.diff(t0[wv, :, 0])
        print(f"{wv:4d} | {med(0, 5):19.0f} | {med(5, 6):30.0f} | {med(6, 3):12.0f} | {med(3, 4):7.0f} | {np.median(period):11.0f}")
    kw_ = int(kw)
    print(f"wave 0, MEAN cycles by tap (step % {kw_}): top->first MFMA | MFMAs | tail+vmcnt | barrier | behind barrier -> next top | period")
    for kx in range(kw_):
        idx = [st for st in range(2, STEPS - 1) if st % kw_ == kx]
        seg = lambda a, b: np.mean([t[0, st, b] - t[0, st, a] for st in idx])
        gap = np.mean([t[0, st + 1, 0] - t[0, st, 4] for st in idx])
        per = np.mean([t[0, st + 1, 0] - t[0, st, 0] for st in idx])
        print(f"  tap {kx}: {seg(0, 5):6.0f} | {seg(5, 6):6.0f} | {seg(6, 3):6.0f} | {seg(3, 4):6.0f} | {gap:6.0f} | {per:6.0f}")
    print("steps 2..9 of wave 0 (G0) and wave 4 (G1), stamps relative to the first:")
    for wv in (0, 3):
        for st in range(2, 10):
            print(f"  wave {wv} step {st}: " + " ".join(f"{int(v - base):7d}" for v in t[wv, st, [0, 5, 6, 3, 4]]))


if __name__ == "__main__":
    main()
