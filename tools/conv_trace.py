#!/usr/bin/env python
"""Per-wavefront timeline of the LDS-DMA conv K loop (s_memtime stamps at the phase boundaries; diagnostic build of the library).

    python tools/conv_trace.py --build                      # here (no GPU needed): lib/libfgt_hip_trace.so with -DFGT_CONV_TRACE
    python tools/conv_trace.py [--layer e20enc10] [--tiles 128x128x8,128x128x8ea]   # on the MI355X

Stamps (cycles, s_memtime): 0 step top | 1 fragments in registers (lgkmcnt 0) | 2 after the stage-release barrier (ea only) | 3 DMAs of the
next tile issued (ea) / fragments in registers (plain) | 4 MFMAs issued | 5 vmcnt wait over | 6 after the closing barrier.
Output: median segment lengths over wavefronts and steps, the step period, and how the two workgroups of a CU sit relative to each other.
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRACE_LIB = os.path.join(ROOT, "fgt_amd", "lib", "libfgt_hip_trace.so")
TR_STEPS, TR_NST, TR_HDR = 32, 8, 12

LAYERS = {  # name: (N, H, W, C0, C1, Cout, groups, k, stride, pad)
    "e20enc10": (20, 60, 108, 256, 384, 512, 2, 3, 1, 1),
    "e20enc8": (20, 60, 108, 256, 0, 384, 1, 3, 1, 1),
    "b8qkv": (1, 1, 97920, 512, 0, 1536, 1, 1, 1, 0),
    "b8ffn1": (1, 1, 97920, 512, 0, 1960, 1, 1, 1, 0),
}
WAVES = {"128x128x8xy": 8, "128x128x8lw": 10, "128x128lw": 6, "128x64lw": 6, "128x128": 4, "128x128ea": 4, "128x128x8": 8, "128x128x8ea": 8, "256x128x16": 16, "256x128x16ea": 16, "256x64x8": 8, "256x64x8ea": 8, "128x64": 4, "128x64ea": 4}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--layer", default="e20enc10")
    ap.add_argument("--tiles", default="128x128x8,128x128x8ea")
    ap.add_argument("--dump", default="")
    ap.add_argument("--warm", type=int, default=30)
    a = ap.parse_args()
    if a.build:
        from fgt_amd import build
        print(build.build(variant="trace", extra_flags=["-DFGT_CONV_TRACE"]))
        return
    os.environ["FGT_HIP_LIB"] = TRACE_LIB
    os.environ["FGT_DIAG_TILES_ALL"] = "1"          # the instrumented kernels live in csrc/diag/conv_split_variants.hip
    import numpy as np
    import torch
    from fgt_amd import _lib, ops

    h = _lib.lib()
    h.fgt_debug_conv_trace.argtypes = [C.c_void_p, C.c_long]
    h.fgt_debug_conv_trace.restype = C.c_int
    dev = torch.device("cuda:0")
    N, H, W, C0, C1, Cout, g, k, s, p = LAYERS[a.layer]
    x = torch.randn(N, H, W, C0, device=dev)
    x1 = torch.randn(N, H, W, C1, device=dev) if C1 else None
    w = torch.randn(Cout, (C0 + C1) // g, k, k, device=dev) * 0.02
    pc = ops.PackedConv(w, torch.zeros(Cout, device=dev), groups=g)
    xs, x1s = ops.split(x), (ops.split(x1) if C1 else None)
    for tile in a.tiles.split(","):
        nw = WAVES[tile]
        per_wg = nw * (TR_HDR + TR_STEPS * TR_NST)
        words = per_wg * 8192
        buf = torch.zeros(words, dtype=torch.int32, device=dev)
        assert h.fgt_debug_conv_trace(C.c_void_p(buf.data_ptr()), C.c_long(words)) == 0
        for _ in range(a.warm):   # warm caches / clocks (sustained load, like the bench); the last launch's trace is the one read
            ops.conv2d(xs, pc, x1=x1s, stride=s, pad=p, act="lrelu", tile=tile, precision="bf16x3")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = ops.conv2d(xs, pc, x1=x1s, stride=s, pad=p, act="lrelu", tile=tile, precision="bf16x3")
        e1.record()
        torch.cuda.synchronize()
        h.fgt_debug_conv_trace(None, 0)
        ms = e0.elapsed_time(e1)
        fl = 2.0 * (out.numel() // Cout) * (Cout // g) * pc.K * g
        raw = buf.cpu().numpy().view(np.uint32).reshape(-1, per_wg)
        if a.dump:
            np.save(f"{a.dump}_{tile}.npy", raw)
        used = raw[:, 3] > 0          # header word nk of wave 0
        raw = raw[used]
        hdr = raw[:, : nw * TR_HDR].reshape(-1, nw, TR_HDR)
        st = raw[:, nw * TR_HDR:].reshape(-1, nw, TR_STEPS, TR_NST).astype(np.int64)
        nk = int(hdr[0, 0, 3])
        steps = min(nk, TR_STEPS)
        print(f"\n== {a.layer} tile {tile}: {ms * 1e3:.0f} us (traced build), {fl / ms / 1e9:.0f} TF alg., {raw.shape[0]} workgroups traced, nk = {nk}")
        h64 = hdr.astype(np.int64)
        pro = (h64[:, :, 4] - h64[:, :, 2]) & 0xFFFFFFFF
        loop = (h64[:, :, 5] - h64[:, :, 4]) & 0xFFFFFFFF
        epi = (h64[:, :, 6] - h64[:, :, 5]) & 0xFFFFFFFF
        epi_issue = (h64[:, :, 8] - h64[:, :, 5]) & 0xFFFFFFFF
        print(f"   epilogue: {np.median(epi_issue):.0f} cycles until the last store is issued (per wavefront, median; p90 {np.percentile(epi_issue, 90):.0f}), {np.median(epi):.0f} until all stores are acknowledged")
        life = ((h64[:, 0, 6] - h64[:, 0, 2]) & 0xFFFFFFFF).astype(np.float64)
        real = h64[:, 0, 7].astype(np.float64)            # 10-ns ticks
        ok = real > 100
        ghz = life[ok] / (real[ok] * 10.0)
        print(f"   shader clock while a workgroup is resident (s_memtime / s_memrealtime): median {np.median(ghz):.2f} GHz (p10 {np.percentile(ghz, 10):.2f}, p90 {np.percentile(ghz, 90):.2f})"
              f" -> the 2.5 PF bf16 peak (2.4 GHz) scales to {2500 * np.median(ghz) / 2.4:.0f} TF; this launch issued {3 * fl / ms / 1e9:.0f} TF = {3 * fl / ms / 1e9 / (2500 * np.median(ghz) / 2.4) * 100:.0f} % of that")
        print(f"   per workgroup (median over wavefronts): prologue {np.median(pro):.0f}  K loop {np.median(loop):.0f} ({np.median(loop) / nk:.0f} per step)  epilogue {np.median(epi):.0f} cycles"
              f"   -> K loop share {np.median(loop) / (np.median(pro) + np.median(loop) + np.median(epi)):.2f}")
        # occupancy of a CU slot over the kernel: workgroups that ran on the same (xcc, se, cu), ordered by start
        lo, hi = 3, steps - 3
        seg_names = ["0>1 reads+lgkm", "1>2 barrier A", "2>3 DMA issue", "3>4 MFMA issue", "4>5 vmcnt wait", "5>6 barrier B"]
        d = (st[:, :, lo:hi, 1:7] - st[:, :, lo:hi, 0:6]) & 0xFFFFFFFF
        period = (st[:, :, lo + 1:hi, 0] - st[:, :, lo:hi - 1, 0]) & 0xFFFFFFFF
        print(f"   step period: median {np.median(period):.0f} cycles (p10 {np.percentile(period, 10):.0f}, p90 {np.percentile(period, 90):.0f})")
        for i, nme in enumerate(seg_names):
            v = d[..., i]
            print(f"   {nme:16s} median {np.median(v):6.0f}  mean {v.mean():7.0f}  p10 {np.percentile(v, 10):6.0f}  p90 {np.percentile(v, 90):6.0f}")
        # co-resident workgroups: same (xcc, se, cu) and overlapping in time
        hw = hdr[:, 0, 0]
        cu = ((hdr[:, 0, 1] & 0xF).astype(np.int64) << 16) | (((hw >> 13) & 0x7).astype(np.int64) << 8) | ((hw >> 8) & 0xF)
        t0 = st[:, 0, 0, 0]
        offs = []
        per = np.median(period)
        for c in np.unique(cu):
            idx = np.nonzero(cu == c)[0]
            for ii in range(len(idx)):
                for jj in range(ii + 1, len(idx)):
                    a_, b_ = idx[ii], idx[jj]
                    # both in their steady state at the same time?
                    sa, sb = st[a_, 0, lo:hi, 3], st[b_, 0, lo:hi, 3]       # start of the MFMA block of every step
                    if sa[-1] < sb[0] or sb[-1] < sa[0]:
                        continue
                    for t in sa:
                        j = np.argmin(np.abs(sb - t))
                        offs.append(((sb[j] - t) % per) / per)
        gaps, spans = [], []
        for c in np.unique(cu)[:64]:
            idx = np.nonzero(cu == c)[0]
            s0 = h64[idx, 0, 2]
            e0_ = h64[idx, 0, 6]
            order = np.argsort((s0 - s0.min()) & 0xFFFFFFFF)
            ss = ((s0 - s0.min()) & 0xFFFFFFFF)[order]
            ee = ((e0_ - s0.min()) & 0xFFFFFFFF)[order]
            spans.append(ee.max())
            busy = np.zeros(int(ee.max()) // 1000 + 2)
            for a_, b_ in zip(ss, ee):
                busy[int(a_) // 1000:int(b_) // 1000 + 1] += 1
            gaps.append(busy[: int(ee.max()) // 1000].mean())
        print(f"   CU view (first 64 CUs): kernel span {np.median(spans):.0f} cycles, mean resident workgroups per CU over the span {np.mean(gaps):.2f}")
        if offs:
            hist, _ = np.histogram(offs, bins=8, range=(0, 1))
            print(f"   phase of the co-resident workgroup's MFMA block within this one's step (8 bins of the period): {hist.tolist()}")
        # one workgroup, all wavefronts, a few steps: absolute stamps relative to wave 0's step top
        wg = min(600, raw.shape[0] - 1)
        print(f"   workgroup {wg} (hw_id {hdr[wg, 0, 0]:#x} xcc {hdr[wg, 0, 1] & 0xF}): stamps of steps {lo}..{lo + 2} relative to wave 0's step top")
        for kt in range(lo, lo + 3):
            base = st[wg, 0, kt, 0]
            for wv in range(nw):
                simd = (hdr[wg, wv, 0] >> 4) & 3
                print(f"     kt {kt} wave {wv} simd {simd}: " + " ".join(f"{int((st[wg, wv, kt, i] - base) & 0xFFFFFFFF):6d}" for i in range(7)))


if __name__ == "__main__":
    main()
