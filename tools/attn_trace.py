#!/usr/bin/env python
"""Per-wavefront timeline of the split-input attention tile loop (s_memtime stamps at the phase boundaries; diagnostic build of the library).
Written at the end of round 2 — seven schedule changes had left the kernel's time unchanged (attn_split_kernel) — and not run yet.

    python tools/attn_trace.py --build                      # here (no GPU needed): lib/libfgt_hip_atrace.so with -DFGT_ATTN_TRACE
    timeout 120 python tools/attn_trace.py [--b 8 --t 17]   # on the MI355X

Stamps (cycles): 0 tile top | 1 own LDS-DMA pieces landed (vmcnt) | 2 behind the barrier | 3 DMAs of a later tile issued | 4 QK^T MFMAs issued
(K fragments waited for) | 5 softmax done (needs the QK^T results) | 6 PV MFMAs issued (V fragments waited for).  Output: median segment
lengths over wavefronts and tiles, the tile period, the shader clock in the kernel, and the stamps of one workgroup's wavefronts side by side.
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRACE_LIB = os.path.join(ROOT, "fgt_amd", "lib", "libfgt_hip_atrace.so")
ATR_TILES, ATR_NST, ATR_HDR = 24, 8, 8
SEG = ["0>1 vmcnt wait", "1>2 barrier", "2>3 DMA issue", "3>4 K reads + QK^T issue", "4>5 softmax (+ QK^T results)", "5>6 V reads + P cvt + PV issue"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--b", type=int, default=8)
    ap.add_argument("--t", type=int, default=17)
    ap.add_argument("--warm", type=int, default=10)
    a = ap.parse_args()
    if a.build:
        from fgt_amd import build
        print(build.build(variant="atrace", extra_flags=["-DFGT_ATTN_TRACE"], swap={"attention_split.hip": "diag/attention_split_trace.hip"}, verbose=False))
        return
    os.environ["FGT_HIP_LIB"] = TRACE_LIB
    import numpy as np
    import torch
    from fgt_amd import _lib, ops

    h = _lib.lib()
    h.fgt_debug_attn_trace.argtypes = [C.c_void_p, C.c_long]
    h.fgt_debug_attn_trace.restype = C.c_int
    dev = torch.device("cuda:0")
    nh, nw, nwv = 20, 36, 8
    qkv = torch.randn(a.b * a.t * nh * nw, 1536, device=dev)
    for fmt in ("f16", "bf16x3"):
        sp = ops.split(qkv, h=(fmt == "f16"))
        per_wg = nwv * (ATR_HDR + ATR_TILES * ATR_NST)
        n_wg = a.b * 16 * -(-(a.t * (nh // 2) * (nw // 2)) // 256)
        words = per_wg * n_wg
        buf = torch.zeros(words, dtype=torch.int32, device=dev)
        for _ in range(a.warm):
            ops.attention_temporal(sp, a.b, a.t, nh, nw, 4, 2, 512)
        assert h.fgt_debug_attn_trace(C.c_void_p(buf.data_ptr()), C.c_long(words)) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.attention_temporal(sp, a.b, a.t, nh, nw, 4, 2, 512)
        e1.record()
        torch.cuda.synchronize()
        h.fgt_debug_attn_trace(None, 0)
        raw = buf.cpu().numpy().view(np.uint32).reshape(-1, per_wg)
        raw = raw[raw[:, 3] > 0]
        hdr = raw[:, : nwv * ATR_HDR].reshape(-1, nwv, ATR_HDR).astype(np.int64)
        st = raw[:, nwv * ATR_HDR:].reshape(-1, nwv, ATR_TILES, ATR_NST).astype(np.int64)
        ntiles = int(hdr[0, 0, 3])
        lo, hi = 4, min(ntiles, ATR_TILES) - 2
        print(f"\n== temporal attention {fmt} b={a.b} t={a.t}: {e0.elapsed_time(e1) * 1e3:.0f} us (traced build), {raw.shape[0]} workgroups traced, {ntiles} tiles each")
        life = ((hdr[:, 0, 4] - hdr[:, 0, 2]) & 0xFFFFFFFF).astype(np.float64)
        real = hdr[:, 0, 5].astype(np.float64)               # 10-ns ticks
        ok = real > 100
        print(f"   shader clock in the kernel (s_memtime / s_memrealtime): median {np.median(life[ok] / (real[ok] * 10.0)):.2f} GHz; "
              f"workgroup lifetime median {np.median(life):.0f} cycles = {np.median(life) / ntiles:.0f} per tile")
        d = (st[:, :, lo:hi, 1:7] - st[:, :, lo:hi, 0:6]) & 0xFFFFFFFF
        period = (st[:, :, lo + 1:hi, 0] - st[:, :, lo:hi - 1, 0]) & 0xFFFFFFFF
        print(f"   tile period: median {np.median(period):.0f} cycles (p10 {np.percentile(period, 10):.0f}, p90 {np.percentile(period, 90):.0f})")
        for i, name in enumerate(SEG):
            v = d[..., i]
            print(f"   {name:34s} median {np.median(v):6.0f}  mean {v.mean():7.0f}  p10 {np.percentile(v, 10):6.0f}  p90 {np.percentile(v, 90):6.0f}")
        wg = min(300, raw.shape[0] - 1)
        print(f"   workgroup {wg} (xcc {hdr[wg, 0, 1] & 0xF}): stamps of tiles {lo}..{lo + 1} relative to wave 0's tile top")
        for it in range(lo, lo + 2):
            base = st[wg, 0, it, 0]
            for wv in range(nwv):
                simd = (hdr[wg, wv, 0] >> 4) & 3
                print(f"     tile {it} wave {wv} simd {simd}: " + " ".join(f"{int((st[wg, wv, it, i] - base) & 0xFFFFFFFF):6d}" for i in range(7)))


if __name__ == "__main__":
    main()
