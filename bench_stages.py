"""BASELINE config C4 and the tool stages around the FGT stage, timed on the MI355X for bench.py's `c4` object.

For one synthetic N-frame H x W clip (default 80 x 240 x 432, the clip of the headline) every covered stage of
tool/video_inpainting.py is run on its full-clip workload through the same entry points the drop-in uses:

    RAFT (calculate_flow :233-288)      fgt_amd.flow_pipeline.compute_flows     2(N-1) pairs, 20 iterations, at H x W and at 2H x 2W
    diffusion fill (:42-51)             fgt_amd.flow_pipeline.diffusion         2 directions x (N-1) flows x 2 channels
    LAFC (complete_flow :342-385)       fgt_amd.flow_pipeline.complete_flows    2 directions x (N-1) flows
    gradient propagation (:623-633)     fgt_amd.propagation.propagate_gradients one clip
    Poisson blend (:644-682)            fgt_amd.blending.poisson_blend_clip     N frames x 3 channels
    FGT stage (:687-748)                the headline of bench.py (passed in)

Each entry carries: ms per unit, the algorithmic work (SURVEY.md §8d counts: reference FLOPs for the networks, bytes for the
solvers / propagation), the roofline it is held against (2.5 PF dense bf16 for the MFMA stages, 8 TB/s HBM for the others) and a CPU
baseline on a BOUNDED sample (the oracle, on rank 0 only).  `rooflines` = per-kernel HIP-event figures over one extra pass of
the flow stages (conv/GEMM TF as executed, and the HBM-bound kernels: corr lookup, warp, small-Cout convs, pointwise), and
`pipeline_frames_per_s` = N / (sum of the stage times) for the whole covered chain.

Not covered (out of scope): image I/O, cv2.resize / cv2.inpaint, scipy.ndimage mask dilation, edge maps.
"""
import json
import os
import time

import numpy as np
import torch

PEAK_BF16 = 2500.0
PEAK_FP32 = 157.3
PEAK_HBM = 8000.0
GF_LAFC = 130.2                     # per LAFC call at 240x432
GF_RAFT = {(240, 432): 245.7, (480, 864): 998.9}


def _sync():
    torch.cuda.synchronize()


def _timed(fn, reps=2, warm=1):
    for _ in range(warm):
        fn()
    _sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    _sync()
    return (time.perf_counter() - t0) / reps, out


def _smooth(x, k):
    """separable box low-pass on the last two dims (torch only: no scipy needed on the product side)"""
    sh = x.shape
    y = x.reshape(-1, 1, sh[-2], sh[-1])
    y = torch.nn.functional.avg_pool2d(y, (k, 1), 1, (k // 2, 0), count_include_pad=False)
    y = torch.nn.functional.avg_pool2d(y, (1, k), 1, (0, k // 2), count_include_pad=False)
    return y.reshape(sh)


def stage_inputs(N, H, W, seed=4321):
    """Seeded synthetic inputs of every stage for one clip (CPU tensors)."""
    g = torch.Generator().manual_seed(seed)
    # video 0..255 with structure at several scales (RAFT needs texture), moving 1.5 px / frame
    base = _smooth(torch.rand(3, H + 64, W + 2 * N + 64, generator=g), 9)
    base = (base - base.amin()) / (base.amax() - base.amin())
    fine = torch.rand(3, H + 64, W + 2 * N + 64, generator=g) * 0.15
    tex = (base + fine).clamp(0, 1)
    video = torch.stack([tex[:, 32 + (i % 3):32 + (i % 3) + H, 16 + 2 * i:16 + 2 * i + W] for i in range(N)]) * 255.0
    # object-like hole: an ellipse drifting 1.5 px / frame (about 17 k px at 240x432), and its dilation
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    hole = torch.stack([(((yy - H / 2 - 0.3 * i) / (H * 0.29)) ** 2 + ((xx - W * 0.35 - 1.5 * i) / (W * 0.185)) ** 2 <= 1.0) for i in range(N)])
    # smooth flows of a few pixels; backward ~ -forward + noise
    ff = _smooth(torch.randn(N - 1, 2, H, W, generator=g), 15) * 40.0
    fb = -ff + _smooth(torch.randn(N - 1, 2, H, W, generator=g), 15) * 4.0
    img = video / 255.0                                                      # [N,3,H,W]
    gx = torch.zeros(N, H, W, 3)
    gy = torch.zeros(N, H, W, 3)
    im = img.permute(0, 2, 3, 1)
    gx[:, :, :-1] = im[:, :, 1:] - im[:, :, :-1]
    gy[:, :-1] = im[:, 1:] - im[:, :-1]
    gx[hole] = 0
    gy[hole] = 0
    return dict(video=video, hole=hole, flow_f=ff, flow_b=fb, gx=gx, gy=gy, img=im.contiguous())


def _models(dev):
    import argparse
    from fgt_amd import lafc_model, raft_model
    from fgt_amd.synth import synth_state_dict
    lafc = lafc_model.Model(dict(lafc_model.DEFAULT_CONFIG)).eval()
    lsd = synth_state_dict(lafc.state_dict(), seed=0, mode="kaiming")
    lafc.load_state_dict(lsd, strict=True)
    raft = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
    rsd = synth_state_dict(raft.state_dict(), seed=0, mode="kaiming")
    raft.load_state_dict(rsd, strict=True)
    return lafc.to(dev), lsd, raft.to(dev), rsd


def _mfma(ms_unit, gflop_unit, prec):
    passes = 3 if prec == "bf16x3" else 1
    peak = PEAK_FP32 if prec == "fp32" else PEAK_BF16
    alg = gflop_unit / ms_unit                                               # GFLOP / ms = TFLOP/s
    return {"bound": "mfma", "algorithmic_tflops_reference_count": round(alg, 2), "achieved": round(alg * passes, 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(alg * passes / peak, 4), "mfma_passes_per_product": passes}


def _counter_traffic(prec):
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "kernel_traffic.json")
    try:
        return json.load(open(p)).get(prec, {})
    except (OSError, ValueError):
        return {}


def _hbm(ms, nbytes):
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM, "unit": "GB/s", "frac": round(gbs / PEAK_HBM, 4), "algorithmic_bytes": int(nbytes)}


def c2_spatial_mhsa(dev, model, prec, t=10, reps=5):
    """BASELINE config C2: ONE spatial window MHSA module (SWMHSA: flow re-weighting Linear + sigmoid, three LayerNorms, global-token
    depthwise pools, q / k / v Linears, window + global-token attention, output Linear; attention_flow.py:57-113) on synthetic tokens of a
    432x240x10 clip (token grid 20x36, x, f ~ N(0,1)).  The north star's "MFMA utilisation in spatial MHSA" is a MODULE figure: 94 % of the
    module's flops are in its Linears, so it is reported as (algorithmic flops of every MFMA launch of the module x MFMA passes) / (module
    wall time) / peak — next to the same figure over the MFMA kernels' own time and the attention kernel's two rooflines."""
    from fgt_amd import ops
    saved = (ops.DEFAULT_CONV_PRECISION, ops.DEFAULT_ATTN_PRECISION)
    ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = prec
    net = model.net
    P = net.packed()
    th, tw = 20, 36
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(t * th * tw, 512, generator=g).to(dev)
    f = torch.randn(t * th * tw, 256, generator=g).to(dev)
    fn = lambda: net._spatial_attention(x, f, P["s0"], t, th, tw)
    dt, _ = _timed(fn, reps=reps, warm=2)
    # ... and as a captured hipGraph (inside a capture the module forks its value / key paths onto two more streams: a call this small is a
    # dependency graph of under-filled launches, and enqueued eagerly it is bound by the host: profiles/r06_run6_*)
    from fgt_amd import fgt_model as _fm
    from fgt_amd.graph import GraphedCall
    variants = {}
    rows_saved = _fm.SPATIAL_STREAM_ROWS
    try:
        gcall = GraphedCall(lambda a, b: net._spatial_attention(a, b, P["s0"], t, th, tw), [x, f])
        gfn = lambda: gcall.graph.replay()
        variants["graph_replay_ms"] = round(_timed(gfn, reps=max(reps, 5), warm=2)[0] * 1e3, 4)
    except Exception as e:  # noqa: BLE001
        variants["error"] = f"{type(e).__name__}: {e}"[:160]
    finally:
        _fm.SPATIAL_STREAM_ROWS = rows_saved
    ops.prof_collect("all")
    ops.prof_enable(True)
    fn()
    _sync()
    ops.prof_enable(False)
    passes = 3 if prec == "bf16x3" else 1
    peak = PEAK_FP32 if prec == "fp32" else PEAK_BF16
    cms, cfl, cn, cby = ops.prof_collect("conv")
    ams, afl, an, aby = ops.prof_collect("attn_spatial")
    ops.prof_collect("all")
    ops.DEFAULT_CONV_PRECISION, ops.DEFAULT_ATTN_PRECISION = saved
    tf = lambda fl, ms: passes * fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return {"workload": f"one SWMHSA module, t = {t}, token grid {th}x{tw} (432x240x{t}), {prec}", "ms_per_module": round(dt * 1e3, 4),
            "algorithmic_gflop": round((cfl + afl) / 1e9, 2), "linears_share_of_flops": round(cfl / max(cfl + afl, 1.0), 3),
            "mfma_frac_module_wall": round(tf(cfl + afl, dt * 1e3) / peak, 4),
            "mfma_frac_mfma_kernels_only": round(tf(cfl + afl, cms + ams) / peak, 4),
            "mfma_frac_linears": round(tf(cfl, cms) / peak, 4), "mfma_frac_attention_kernel": round(tf(afl, ams) / peak, 4),
            "hbm_frac_attention_kernel": round(aby / max(ams * 1e-3, 1e-12) / 1e9 / PEAK_HBM, 4),
            "launches": {"linear_gemm": cn, "attention": an}, "peak_tflops": peak,
            "streams_in_graph_replay": 3 if 0 < t * th * tw <= _fm.SPATIAL_STREAM_ROWS else 1, **variants,
            **({"mfma_frac_module_wall_graph_replay": round(tf(cfl + afl, variants["graph_replay_ms"]) / peak, 4)} if "graph_replay_ms" in variants else {})}


def run_stages(dev, prec="bf16x3", frames=80, H=240, W=432, fgt_ms=None, with_cpu=True, reps=2, fill_iters=None, blend_iters=None, fgt_model=None):
    from fgt_amd import blending, flow_pipeline, ops, propagation
    torch.set_grad_enabled(False)
    saved = (ops.DEFAULT_CONV_PRECISION, ops.DEFAULT_ATTN_PRECISION)
    ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = prec
    N = frames
    inp = stage_inputs(N, H, W)
    lafc, lsd, raft, rsd = _models(dev)
    out = {"clip": f"{N} frames {W}x{H}", "precision": prec, "stages": {}}
    if fgt_model is not None:
        try:
            out["c2_spatial_mhsa"] = c2_spatial_mhsa(dev, fgt_model, prec)
            if prec != "fp32":
                out["c2_spatial_mhsa"]["fp32_exact"] = {k: v for k, v in c2_spatial_mhsa(dev, fgt_model, "fp32").items() if k.startswith(("mfma_frac", "ms_per", "hbm_frac"))}
            # the same module at the size the benchmark's step actually launches it (8 windows x 17 frames = 136 frames per call)
            out["c2_spatial_mhsa"]["at_step_size_t136"] = {k: v for k, v in c2_spatial_mhsa(dev, fgt_model, prec, t=136, reps=3).items()
                                                            if k.startswith(("mfma_frac", "ms_per", "hbm_frac"))}
        except Exception as e:  # noqa: BLE001
            out["c2_spatial_mhsa"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    st = out["stages"]
    times_ms = {}

    # ------------------------------------------------------------------ RAFT, 2(N-1) pairs, 20 iterations
    video = inp["video"].to(dev)
    for scale in (1, 2):
        h, w = H * scale, W * scale
        v = video if scale == 1 else torch.nn.functional.interpolate(video, size=(h, w), mode="bilinear", align_corners=False)
        dt, _ = _timed(lambda: flow_pipeline.compute_flows(raft, v, iters=20), reps=reps)
        pairs = 2 * (N - 1)
        ms_pair = dt * 1e3 / pairs
        gf = GF_RAFT.get((h, w))
        rec = {"ms_per_pair": round(ms_pair, 3), "pairs": pairs, "ms_per_clip": round(dt * 1e3, 1), "iters": 20,
               "pipeline": f"per-frame fnet/cnet cache + {flow_pipeline.RAFT_PAIR_BATCH} pairs per batch (bit-identical flows)"}
        if gf:
            rec["roofline"] = _mfma(ms_pair, gf, prec)
        st[f"raft_{w}x{h}"] = rec
        times_ms[f"raft_{w}x{h}"] = dt * 1e3
    del v
    # ------------------------------------------------------------------ diffusion fill: both directions in two calls like the tool
    hole = inp["hole"].to(dev)
    ffl = inp["flow_f"].to(dev).permute(1, 0, 2, 3)[None].contiguous()         # [1,2,N-1,H,W]
    fbl = inp["flow_b"].to(dev).permute(1, 0, 2, 3)[None].contiguous()
    mk_f = hole[:-1].float()[None, None]                                       # forward flows use mask i, backward mask i+1 (:350-353)
    mk_b = hole[1:].float()[None, None]
    kw = {} if fill_iters is None else {"iters": fill_iters}
    dt, dif_f = _timed(lambda: flow_pipeline.diffusion(ffl, mk_f, **kw), reps=reps)
    dt2, dif_b = _timed(lambda: flow_pipeline.diffusion(fbl, mk_b, **kw), reps=reps)
    hole_px = float(hole.float().sum(dim=(1, 2)).mean())

    def solver_info(name, per_iter_bytes):
        """what the last call of the stage ran: on-chip (one workgroup per problem) or multi-launch CG, iterations used"""
        st = ops.last_solver.get(name, {})
        if st.get("solver") == "onchip":
            it = (st["status"].cpu().long() >> 1).float()
            return {"solver": "on-chip CG: one workgroup per problem, all iterations in one launch (csrc/solve_onchip.hip)", "iterations_mean": round(float(it.mean()), 1),
                    "iterations_max": int(it.max()), "bbox_rows_cols": list(st.get("bbox", ())), "bytes_per_hole_px_iter": 8}
        return {"solver": "multi-launch CG (two launches per iteration over the frame)", "bytes_per_hole_px_iter": per_iter_bytes}

    fill_info = solver_info("laplace_fill", 17 * 4)
    fill_info.setdefault("iterations_mean", flow_pipeline.FILL_ITERS)
    # algorithmic bytes: per iteration and hole pixel what the solver moves through the memory system (multi-launch: ~17 floats; on-chip: the
    # in-place x update only — p, r, A p never leave the CU), plus every map in and out once
    fill_bytes = 2 * (N - 1) * (hole_px * fill_info["iterations_mean"] * fill_info["bytes_per_hole_px_iter"] + H * W * 8)
    st["diffusion_fill"] = {"ms_per_clip": round((dt + dt2) * 1e3, 2), "ms_per_direction": round((dt + dt2) * 5e2, 2), "maps": 4 * (N - 1),
                            "hole_px_per_map": int(hole_px), "solver": fill_info, "roofline": _hbm((dt + dt2) * 5e2, fill_bytes)}
    times_ms["diffusion_fill"] = (dt + dt2) * 1e3
    # ------------------------------------------------------------------ LAFC over both directions
    dtl, comp_f = _timed(lambda: flow_pipeline.complete_flows(lafc, ffl, mk_f, dif_f), reps=reps)
    dtl2, comp_b = _timed(lambda: flow_pipeline.complete_flows(lafc, fbl, mk_b, dif_b), reps=reps)
    ms_flow = (dtl + dtl2) * 1e3 / (2 * (N - 1))
    st["lafc"] = {"ms_per_flow": round(ms_flow, 3), "flows": 2 * (N - 1), "ms_per_clip": round((dtl + dtl2) * 1e3, 1),
                  "pipeline": f"{flow_pipeline.LAFC_PIVOT_BATCH} pivots per call",
                  "roofline": _mfma(ms_flow, GF_LAFC * (H * W) / (240 * 432), prec)}
    times_ms["lafc"] = (dtl + dtl2) * 1e3
    # ------------------------------------------------------------------ gradient propagation
    gx, gy = inp["gx"].to(dev), inp["gy"].to(dev)
    flf = comp_f.permute(0, 2, 3, 1).contiguous()                               # [N-1,H,W,2]
    flb = comp_b.permute(0, 2, 3, 1).contiguous()
    dtp, (pgx, pgy, tofill) = _timed(lambda: propagation.propagate_gradients(gx, gy, hole, flf, flb), reps=reps)
    prop_bytes = N * H * W * (2 * 3 * 4 * 2 + 1 + 1) + 2 * (N - 1) * H * W * 2 * 4
    st["gradient_propagation"] = {"ms_per_clip": round(dtp * 1e3, 3), "hole_px_total": int(hole.sum()),
                                  "roofline": dict(_hbm(dtp * 1e3, prop_bytes), note="gradients in + out, masks, both flow fields once; the stage is a "
                                                   "chain of per-frame launches (latency-bound), not a streaming pass")}
    times_ms["gradient_propagation"] = dtp * 1e3
    # ------------------------------------------------------------------ Poisson blend
    img = inp["img"].to(dev)
    trg = img * (~hole)[..., None]
    kwb = {} if blend_iters is None else {"iters": blend_iters}
    dtb, (blend, unf) = _timed(lambda: blending.poisson_blend_clip(trg, pgx, pgy, hole, tofill, **kwb), reps=reps)
    blend_info = solver_info("poisson_blend", 20 * 4)
    blend_info.setdefault("iterations_mean", blending.BLEND_ITERS)
    blend_bytes = N * 3 * (hole_px * blend_info["iterations_mean"] * blend_info["bytes_per_hole_px_iter"] + H * W * 16)
    st["poisson_blend"] = {"ms_per_clip": round(dtb * 1e3, 2), "problems": 3 * N, "hole_px_per_frame": int(hole_px), "solver": blend_info,
                           "roofline": _hbm(dtb * 1e3, blend_bytes)}
    times_ms["poisson_blend"] = dtb * 1e3
    # the same figure `run_sharded` prints at N > 1: equal checksums = the sharded chain reproduced this one
    out["checksum"] = round(float(blend.double().mean()) + float(comp_f.double().abs().mean()), 6)
    if fgt_ms is not None:
        st["fgt"] = {"ms_per_clip": round(fgt_ms, 2), "note": "the headline of this line (one clip pass)"}
        times_ms["fgt"] = fgt_ms
    # ------------------------------------------------------------------ the covered chain
    chain = ["raft_%dx%d" % (2 * W, 2 * H), "diffusion_fill", "lafc", "gradient_propagation", "poisson_blend"] + (["fgt"] if fgt_ms is not None else [])
    total = sum(times_ms[k] for k in chain)
    out["pipeline_frames_per_s"] = {"value": round(N / (total * 1e-3), 2), "unit": "frames/s", "ms_per_clip": round(total, 1),
                                    "stages_ms": {k: round(times_ms[k], 2) for k in chain},
                                    "note": f"covered chain for one {N}-frame {W}x{H} clip with RAFT on the 2x input ({2 * W}x{2 * H}: the tool-faithful "
                                            "size, SURVEY.md §8a a11); stages run back to back on one GPU, inputs resident in HBM"}
    # ------------------------------------------------------------------ per-kernel HIP-event figures over one pass of the flow stages
    # RAFT and LAFC are profiled separately so that each stage carries BOTH fractions:
    #   frac_effective = reference FLOP count of the stage (includes the fnet / cnet passes the per-frame cache no longer executes) x MFMA
    #                    passes / stage wall time — what a user of the reference gets;
    #   frac_hardware  = FLOPs of the conv / GEMM launches actually executed x MFMA passes / their HIP-event time — what the kernel does.
    passes = 3 if prec == "bf16x3" else 1
    peak = PEAK_FP32 if prec == "fp32" else PEAK_BF16
    rl = []

    def conv_entry(label):
        ms, fl, n, by = ops.prof_collect("conv")
        if n == 0 or ms <= 0:
            return None
        return {"kind": label, "bound": "mfma", "achieved": round(passes * fl / ms / 1e9, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(passes * fl / ms / 1e9 / peak, 4), "algorithmic_tflops": round(fl / ms / 1e9, 2), "launches": n, "avg_launch_us": round(1e3 * ms / n, 2)}

    def relabel(stage_keys, hw):
        for k in stage_keys:
            r = st.get(k, {}).get("roofline")
            if r and hw:
                r["frac_effective"] = r["frac"]
                r["frac_hardware"] = hw["frac"]
                r["note"] = ("frac = frac_effective: reference FLOP count x MFMA passes / stage time; frac_hardware: executed conv/GEMM launches, HIP-event timed "
                             "(864x480 sample for RAFT)" if k.startswith("raft") else "frac = frac_effective; frac_hardware: executed conv launches, HIP-event timed")

    ops.prof_collect("all")
    ops.prof_enable(True)
    v2 = torch.nn.functional.interpolate(video[:17], size=(2 * H, 2 * W), mode="bilinear", align_corners=False)
    flow_pipeline.compute_flows(raft, v2, iters=20)
    _sync()
    hw_raft = conv_entry(f"conv (RAFT 16 frames at {2 * W}x{2 * H}, as executed)")
    del v2
    other = {}
    for kind in ("corr_lookup", "conv_small", "pointwise"):
        other[kind] = ops.prof_collect(kind)
    flow_pipeline.complete_flows(lafc, ffl[:, :, :16], mk_f[:, :, :16], dif_f[:, :, :16])
    _sync()
    hw_lafc = conv_entry("conv (LAFC 16 flows, as executed)")
    for kind in ("conv_small", "pointwise"):
        a, b = other[kind], ops.prof_collect(kind)
        other[kind] = tuple(x + y for x, y in zip(a, b))
    # the north star's "bandwidth-bound warp": fbConsistencyCheck's two image_warp calls over every flow pair of the clip
    # (LAFC/models/utils/fbConsistencyCheck.py:33-35: 2-channel H x W maps, N-1 pairs per call) and one 3-channel frame warp of the clip
    ops.warp(flf, flb)
    ops.warp(flb, flf)
    ops.warp(img[: N - 1].contiguous(), flf)
    _sync()
    other["warp"] = ops.prof_collect("warp")
    ops.prof_enable(False)
    for e in (hw_raft, hw_lafc):
        if e:
            rl.append(e)
    relabel([k for k in st if k.startswith("raft")], hw_raft)
    relabel(["lafc"], hw_lafc)
    for kind in ("corr_lookup", "conv_small", "warp", "pointwise"):
        ms, fl, n, by = other[kind]
        if n == 0 or ms <= 0:
            continue
        e = dict(_hbm(ms, by), kind=kind, launches=n, avg_launch_us=round(1e3 * ms / n, 2))
        tr = _counter_traffic(prec).get(kind)
        if tr and tr.get("traffic_over_algorithmic"):
            # rocprofv3 FETCH_SIZE / WRITE_SIZE of this kernel at the same shapes (tools/hbm_micro.py under --pmc, profiles/kernel_traffic.json)
            e.update(traffic_over_algorithmic=tr["traffic_over_algorithmic"], counter_bytes_per_launch=tr.get("hbm_bytes_per_launch"),
                     counter_GBps=round(e["achieved"] * tr["traffic_over_algorithmic"], 1))
        rl.append(e)
    ops.prof_collect("all")
    out["rooflines"] = rl
    # ------------------------------------------------------------------ CPU baselines on bounded samples (oracle = port of the reference)
    if with_cpu:
        out["cpu_baseline"] = _cpu_legs(inp, lsd, rsd, N, H, W, comp_f, comp_b, st)
    ops.DEFAULT_CONV_PRECISION, ops.DEFAULT_ATTN_PRECISION = saved
    return out


def _cpu_legs(inp, lsd, rsd, N, H, W, comp_f, comp_b, st):
    """Oracle timings on the host cores; every sample is bounded to a few seconds and says what it was."""
    from fgt_amd import lafc_model
    from oracle import blend_oracle as BO
    from oracle import fill_oracle as FO
    from oracle import lafc_oracle as LO
    from oracle import prop_oracle as PO
    from oracle import raft_oracle as RO
    ncpu = os.cpu_count() or 1
    th = min(ncpu, 32)
    torch.set_num_threads(th)
    cpu = {"cores": th, "host_cores": ncpu, "kind": "port"}
    hole = inp["hole"]
    # LAFC: one call (3 flows)
    fl = inp["flow_f"][:3].permute(1, 0, 2, 3)[None].contiguous()
    mk = hole[:3].float()[None, None]
    t0 = time.perf_counter()
    LO.lafc_forward(lsd, lafc_model.DEFAULT_CONFIG, fl * (1 - mk), mk)
    dt = time.perf_counter() - t0
    cpu["lafc"] = {"ms_per_flow": round(dt * 1e3, 1), "sample": "one oracle lafc_forward call (3 flows -> 1 completed flow)",
                   "gpu_speedup": round(dt * 1e3 / st["lafc"]["ms_per_flow"], 1)}
    # RAFT: one pair at H x W, 20 iterations (the 2x input costs ~4x)
    v = inp["video"]
    t0 = time.perf_counter()
    RO.raft_forward(rsd, v[0:1], v[1:2], iters=20)
    dt = time.perf_counter() - t0
    key = f"raft_{W}x{H}"
    cpu[key] = {"ms_per_pair": round(dt * 1e3, 1), "sample": f"one oracle raft_forward pair at {W}x{H}, 20 iterations",
                "gpu_speedup": round(dt * 1e3 / st[key]["ms_per_pair"], 1)}
    # diffusion fill: 2 maps by scipy spsolve (what the reference runs)
    t0 = time.perf_counter()
    for c in range(2):
        FO.regionfill(inp["flow_f"][0, c].numpy(), hole[0].numpy())
    dt = (time.perf_counter() - t0) / 2
    cpu["diffusion_fill"] = {"ms_per_map": round(dt * 1e3, 1), "sample": "oracle regionfill (scipy spsolve) on 2 maps",
                             "ms_per_clip_extrapolated": round(dt * 1e3 * 4 * (N - 1), 0),
                             "gpu_speedup": round(dt * 1e3 * 4 * (N - 1) / st["diffusion_fill"]["ms_per_clip"], 1)}
    # propagation: the first 8 frames (vectorised numpy port; the reference's own per-pixel loop needs minutes)
    n8 = min(8, N)
    a = lambda t: np.ascontiguousarray(t.numpy())
    t0 = time.perf_counter()
    PO.get_flownn_gradient(a(inp["gx"][:n8]), a(inp["gy"][:n8]), a(hole[:n8]), a(comp_f[: n8 - 1].permute(0, 2, 3, 1).cpu()), a(comp_b[: n8 - 1].permute(0, 2, 3, 1).cpu()))
    dt = time.perf_counter() - t0
    cpu["gradient_propagation"] = {"ms_per_clip_extrapolated": round(dt * 1e3 * N / n8, 0), "sample": f"oracle get_flownn_gradient (vectorised numpy) on the first {n8} frames, scaled by frames",
                                   "gpu_speedup": round(dt * 1e3 * N / n8 / st["gradient_propagation"]["ms_per_clip"], 1)}
    # Poisson blend: one frame, one channel-set by scipy LSQR (the reference's solver)
    im = a(inp["img"][0])
    h0 = a(hole[0])
    trg = im * (~h0)[..., None]
    gx = a(inp["gx"][0])[:, : W - 1]
    gy = a(inp["gy"][0])[: H - 1]
    t0 = time.perf_counter()
    BO.poisson_blend(trg, gx, gy, h0, np.zeros_like(h0), tight=False)
    dt = time.perf_counter() - t0
    cpu["poisson_blend"] = {"ms_per_frame": round(dt * 1e3, 1), "sample": "oracle poisson_blend (scipy LSQR at the reference's tolerances) on one frame (3 channels)",
                            "ms_per_clip_extrapolated": round(dt * 1e3 * N, 0), "gpu_speedup": round(dt * 1e3 * N / st["poisson_blend"]["ms_per_clip"], 1)}
    return cpu


def run_sharded(dev, prec, rank, world, group=None, frames=80, H=240, W=432, fgt_ms=None, backend="nccl"):
    """N > 1: the covered chain with every stage's units block-sharded over the ranks (fgt_amd.flow_pipeline / blending: RAFT pairs, fill
    maps, LAFC pivots, Poisson frames; one all-gather per stage; the latency-bound gradient propagation is replicated on every rank) —
    one warm pass (weight packing, tile tuning), then one timed pass per stage between barriers, max over ranks.  Returns on every rank
    {stage: ms, ..., "pipeline_frames_per_s"} (strong scaling of the same 80-frame clip as `c4`)."""
    import torch.distributed as dist
    from fgt_amd import blending, flow_pipeline, ops, propagation
    torch.set_grad_enabled(False)
    saved = (ops.DEFAULT_CONV_PRECISION, ops.DEFAULT_ATTN_PRECISION)
    ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = prec
    N = frames
    inp = stage_inputs(N, H, W)
    lafc, lsd, raft, rsd = _models(dev)
    kw = dict(rank=rank, world=world, group=group)

    def barrier():
        torch.cuda.synchronize()
        dist.barrier(group=group)
        torch.cuda.synchronize()

    def timed(fn):
        fn()                                    # warm pass
        barrier()
        t0 = time.perf_counter()
        out = fn()
        barrier()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)
        return tt.item() * 1e3, out

    ms = {}
    video = inp["video"].to(dev)
    v2 = torch.nn.functional.interpolate(video, size=(2 * H, 2 * W), mode="bilinear", align_corners=False)
    ms[f"raft_{2 * W}x{2 * H}"], _ = timed(lambda: flow_pipeline.compute_flows(raft, v2, iters=20, **kw))
    del v2
    hole = inp["hole"].to(dev)
    ffl = inp["flow_f"].to(dev).permute(1, 0, 2, 3)[None].contiguous()
    fbl = inp["flow_b"].to(dev).permute(1, 0, 2, 3)[None].contiguous()
    mk_f, mk_b = hole[:-1].float()[None, None], hole[1:].float()[None, None]
    bounds = ops.hole_bounds(hole)                                           # one read-back per clip: both fills and the blend use these holes
    t1, dif_f = timed(lambda: flow_pipeline.diffusion(ffl, mk_f, bounds=bounds, **kw))
    t2, dif_b = timed(lambda: flow_pipeline.diffusion(fbl, mk_b, bounds=bounds, **kw))
    ms["diffusion_fill"] = t1 + t2
    t1, comp_f = timed(lambda: flow_pipeline.complete_flows(lafc, ffl, mk_f, dif_f, **kw))
    t2, comp_b = timed(lambda: flow_pipeline.complete_flows(lafc, fbl, mk_b, dif_b, **kw))
    ms["lafc"] = t1 + t2
    gx, gy = inp["gx"].to(dev), inp["gy"].to(dev)
    flf, flb = comp_f.permute(0, 2, 3, 1).contiguous(), comp_b.permute(0, 2, 3, 1).contiguous()
    ms["gradient_propagation"], (pgx, pgy, tofill) = timed(lambda: propagation.propagate_gradients(gx, gy, hole, flf, flb))
    img = inp["img"].to(dev)
    trg = img * (~hole)[..., None]
    ms["poisson_blend"], (blend, unf) = timed(lambda: blending.poisson_blend_clip(trg, pgx, pgy, hole, tofill, bounds=bounds, **kw))
    out = {"stages_ms": {k: round(v, 2) for k, v in ms.items()}, "n_gpus": world,
           "checksum": round(float(blend.double().mean()) + float(comp_f.double().abs().mean()), 6)}
    total = sum(ms.values())
    if fgt_ms is not None:
        out["stages_ms"]["fgt"] = round(fgt_ms, 2)
        total += fgt_ms
    out["ms_per_clip"] = round(total, 1)
    out["pipeline_frames_per_s"] = round(N / (total * 1e-3), 2)
    ops.DEFAULT_CONV_PRECISION, ops.DEFAULT_ATTN_PRECISION = saved
    return out
