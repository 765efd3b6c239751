#!/usr/bin/env python
"""Headline benchmark: inpainted frames/sec of the FGT stage on a synthetic 432x240x80 clip (BASELINE.json).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over the clip: all 16 sliding-window Model.forward calls of the reference
schedule (t = 13,17,18,... sum 275) + uint8 compose + ordered blend, inputs resident in HBM.

N = 1: the headline runs in the bf16x3 arithmetic (fp32 operands split into hi/lo bf16, 3 bf16 MFMAs per product, fp32
accumulate; inside the north star's 1e-3 bar, parity block in the line); the SAME invocation then times the exact-fp32 mode and
reports it as `fp32_exact` (value, ms_per_step, its own roofline against the 157.3 TF fp32-MFMA peak).
N > 1 (`--scaling strong`, default): ONE clip sharded over the ranks — frames block-sharded for the per-frame stages with
chunked RCCL all-to-alls of the feature rows each rank needs, windows cost-balanced over the ranks with equal lengths co-located, window outputs
exchanged as uint8 in one all-gather, identical composite on every rank.  The weak variant (one independent clip
per rank, no data-path collective) is timed first and reported beside it as `weak_scaling_clip_per_rank`; if the sharded
section fails or times out the weak number becomes the headline and the line says so (`scaling`, `strong_error`).

Prints ONE compact JSON line (< 6 KB, `compact_line`) as the LAST stdout line on rank 0 (contract in the task statement); the full
record (per-mode roofline lists, traffic sources, solver details, notes) goes to gpurun_out/bench_detail.json.  `roofline` is the
kernel with the largest share of the step (per-launch HIP events on the launch stream over the timed region); `rooflines` lists the
MFMA kernels of the path (conv/GEMM, temporal attention, spatial attention) and the HBM-bound ones; `cpu_baseline` is the reference
module itself when /root/reference is mounted (kind "reference"), else the oracle (PyTorch CPU restatement, kind "port"), timed on the
host cores on a bounded sample, median of 3 runs.
"""
import argparse
import json
import os
import subprocess
import statistics
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense peak
PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured with a float4 copy)
KERNELS = {"conv": "conv_{taps,wide,split}_kernel (bf16x3) / conv_igemm_kernel (fp32 inputs) / conv_f16_kernel (f16): implicit-GEMM conv + every Linear; all fgt_conv2d MFMA launches",
           "attn_temporal": "attn_split_kernel<8|4, H> (bf16x3: split q/k/v; f16: H = true) / attn_kernel<4> (fp32): temporal zone attention, fgt_attention mode 0",
           "attn_spatial": "attn_split_kernel<2, H> (bf16x3 / f16) / attn_kernel<2> (fp32): spatial window + global-token attention, fgt_attention mode 1"}
# HBM-bound kernels of the step (fgt_prof kinds 3..9): algorithmic bytes (every input / output byte once) over HIP-event time
HBM_KERNELS = {"layernorm": "layernorm_kernel: row LayerNorm over [x0 | x1], up to two affine outputs (fp32 / split / fp16), FGT/models/model.py:126-128,147",
               "fold": "fold_kernel: overlap-add of token patches as a gather (+ 1/count, residual, ReLU, split / fp16 output), ffn_base.py:56-75, model.py:102-110",
               "conv_small": "conv3x3_tiled_kernel<3>: decoder.final 64 -> 3 (+ tanh) on the VALU, model.py:186",
               "dw_pool": "dw_pool4_kernel: depthwise 4x4 / stride 4 global-token extraction, attention_flow.py:44-48",
               "pointwise": "gather_rows / dw3x3_res / split / pad_tokens / pack_frames / nchw<->nhwc / axpby: one pass over the data"}


DTYPES = {"fp32": "f32", "bf16x3": "f32 (conv/GEMM/attention products as 3 bf16 MFMAs on hi/lo splits, fp32 accumulate)",
          "f16": "f16 operands (activations rounded once by their producer, 11 significant bits), one fp16 MFMA per product, fp32 accumulate; "
                 "norms / softmax / residual stream / blends in fp32"}


def fgt_flops(t):
    """Algorithmic FLOPs of one reference Model.forward at 240x432 (torch FlopCounterMode)."""
    return (147.03 * t + 1.0618 * t * t) * 1e9


def kernel_traffic(prec):
    """HBM bytes per launch of the MFMA kernels from the rocprofv3 PMC passes over this same command with the tile table
    pre-seeded (FETCH_SIZE and WRITE_SIZE in separate runs; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM), committed as
    profiles/kernel_traffic.json by tools/gpu_check.sh + tools/pmc_traffic.py.  PMC counters cannot be read from inside the
    process, so this is the last profiled value of this configuration (file carries the commit and command it came from)."""
    p = os.path.join(ROOT, "profiles", "kernel_traffic.json")
    if not os.path.exists(p):
        return {}
    out = dict(json.load(open(p)).get(prec, {}))
    out["_source"] = {"file": "profiles/kernel_traffic.json", "git_head": out.get("git_head", ""), "command": out.get("command", ""),
                      "note": "PMC counters cannot be read in-process: last profiled value of this configuration (tools/gpu_check.sh pmc stage)"}
    # the profile's kernel sources against the tree that runs (a hash over csrc/ + the ABI header, fgt_amd.build.csrc_hash): a kernel change after the PMC
    # visit makes the traffic figure stale, and the line says so instead of quoting it silently; commits that touch no kernel do not
    here = ""
    try:
        from fgt_amd.build import csrc_hash
        here = csrc_hash()
    except Exception:  # noqa: BLE001
        pass
    out["_source"]["tree"] = here
    prof = out.get("csrc_hash", "")
    if here and here != prof:
        out["_source"]["warning"] = f"traffic profiled on kernel sources {prof or '(unrecorded: ' + out['_source']['git_head'] + ')'}, this tree's are {here}: re-run tools/gpu_check.sh pmc"
    return out


def cpu_baseline(cfg, sd, frames, flows, masks, sched, model=None, runs=3, prefer="reference"):
    """CPU path on a bounded sample of the same workload: the first window (t = 13) of the schedule, `runs` timed runs (median)
    after choosing the intra-op thread count that runs a 2-frame probe fastest (a 256-thread pool on a 256-core host is ~8x slower
    than 32 threads for these conv sizes).  The timed function is the REFERENCE module itself (FGT.models.model.Model loaded
    through oracle/reference_loader.py, kind "reference") when /root/reference is mounted — the authoring container; it does not
    exist on the GPU box — and the oracle (PyTorch-CPU restatement pinned to it, kind "port") otherwise.  Parity is always taken
    against the oracle's output (identical to the reference's within fp32 rounding: tests/test_oracle_pinned.py)."""
    from oracle import fgt_oracle as O
    from oracle import reference_loader as RL
    nb, ref = sched[0]
    ids = nb + ref
    m = masks[:, ids].cpu()
    mf = (frames[:, ids].cpu() * 2 - 1) * (1 - m)
    fl = flows[:, ids].cpu()
    kind, run = "port", (lambda a, b, c: O.fgt_forward(sd, cfg, a, b, c))
    if RL.available() and prefer == "reference":
        try:
            refm = RL.fgt_model(dict(cfg))
            refm.load_state_dict({k: v.cpu() for k, v in sd.items()}, strict=True)
            kind, run = "reference", (lambda a, b, c: refm(a, b, c))
        except Exception:  # noqa: BLE001 - fall back to the port, the line says which one ran
            kind = "port"
    ncpu = os.cpu_count() or 1
    best, best_dt = 1, None
    for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(th)
        run(mf[:, :1], fl[:, :1], m[:, :1])
        t0 = time.perf_counter()
        run(mf[:, :2], fl[:, :2], m[:, :2])
        dt = time.perf_counter() - t0
        if best_dt is None or dt < best_dt:
            best, best_dt = th, dt
    torch.set_num_threads(best)
    times = []
    for _ in range(runs):
        t0 = time.perf_counter()
        run(mf, fl, m)
        times.append(time.perf_counter() - t0)
    ref = O.fgt_forward(sd, cfg, mf, fl, m)
    dt = statistics.median(times)
    total = sum(fgt_flops(len(a) + len(b)) for a, b in sched)
    est_clip_s = dt * total / fgt_flops(len(ids))

    def parity_of(model, what="headline precision"):
        """"PSNR vs ref" of the metric: the same window through the HIP path (in the arithmetic mode selected right now) vs the oracle's output"""
        dev = frames.device
        got = model(mf.to(dev), fl.to(dev), m.to(dev)).cpu()
        u8 = lambda x: ((x + 1) / 2 * 255).clamp(0, 255).to(torch.uint8).float()
        return {"window": 0, "frames": len(ids), "max_abs_diff": float((got - ref).abs().max()),
                "ref_max_abs": float(ref.abs().max()), "psnr_db_uint8": round(O.psnr(u8(got), u8(ref)), 2),
                "note": f"HIP path ({what}) vs CPU oracle on window 0; PSNR per FGT/metrics/psnr.py:5-9 on clip((x+1)/2*255) uint8 frames (100 = identical)"}

    parity = parity_of(model) if model is not None else None
    who = "reference FGT.models.model.Model.forward" if kind == "reference" else "oracle fgt_forward (CPU restatement of the reference)"
    return parity, parity_of, {"value": round(frames.shape[1] / est_clip_s, 4), "unit": "frames/s", "cores": best, "host_cores": ncpu,
                    "kind": kind, "runs_s": [round(x, 3) for x in times],
                    "sample": f"{who} on window 0 (t={len(ids)}) at {frames.shape[-1]}x{frames.shape[-2]}: median of {runs} runs = {dt:.2f} s "
                              f"with {best} threads (fastest of a 2-frame probe); clip time extrapolated by F(t)=147.03t+1.0618t^2 GFLOP over the "
                              f"{len(sched)}-window reference schedule ({total / 1e12:.1f} TFLOP)"}


REFERENCE_RECORD = os.path.join("profiles", "cpu_baseline_reference.json")


def cpu_baseline_only(args):
    """`python bench.py --cpu-baseline-only` (no GPU needed; the authoring container, where /root/reference is mounted): time the REAL reference
    module (FGT/models/model.py:12-25, kind "reference") and the oracle port on window 0 of the same synthetic clip with the same thread
    search, check that their outputs agree, and write profiles/cpu_baseline_reference.json.  The GPU box, which has no /root/reference, times the
    port in its own run and quotes this record beside it (`cpu_baseline.reference_record`)."""
    from fgt_amd.fgt_model import DEFAULT_CONFIG
    from fgt_amd.scheduler import window_schedule
    from fgt_amd.synth import synth_clip, synth_state_dict
    from oracle import reference_loader as RL
    torch.set_grad_enabled(False)
    if not RL.available():
        raise SystemExit("--cpu-baseline-only needs the reference tree (FGT_REFERENCE, default /root/reference)")
    cfg = dict(DEFAULT_CONFIG, input_resolution=(args.height, args.width))
    refm = RL.fgt_model(dict(cfg))
    sd = synth_state_dict(refm.state_dict(), seed=0)
    fr, fl, ms = synth_clip(args.frames, args.height, args.width, seed=1234, device=torch.device("cpu"))
    sched = window_schedule(args.frames, 5, 10, -1)
    rec = {"command": "python bench.py --cpu-baseline-only", "clip": f"{args.width}x{args.height}x{args.frames}", "host": os.uname().nodename}
    for kind in ("reference", "port"):
        _, _, cb = cpu_baseline(cfg, sd, fr, fl, ms, sched, model=None, prefer=kind)
        assert cb["kind"] == kind, f"asked for the {kind} baseline, got {cb['kind']}"
        rec[kind] = cb
    from oracle import fgt_oracle as O
    nb, ref = sched[0]
    ids = (nb + ref)[:3]
    m = ms[:, ids]
    mf = (fr[:, ids] * 2 - 1) * (1 - m)
    refm.load_state_dict(sd, strict=True)
    d = (refm(mf, fl[:, ids], m) - O.fgt_forward(sd, cfg, mf, fl[:, ids], m)).abs().max().item()
    rec["max_abs_reference_minus_port"] = d
    rec["port_equals_reference"] = "tests/test_oracle_pinned.py (live against the reference) + the 3-frame check of this record"
    with open(os.path.join(ROOT, REFERENCE_RECORD), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({"cpu_baseline": rec["reference"], "port": {k: rec["port"][k] for k in ("value", "cores", "runs_s")},
                      "max_abs_reference_minus_port": d, "written": REFERENCE_RECORD}))


def reference_record():
    """The committed reference-kind CPU baseline (cpu_baseline_only), for the GPU box's line: value, cores, where it was measured."""
    p = os.path.join(ROOT, REFERENCE_RECORD)
    if not os.path.exists(p):
        return None
    r = json.load(open(p))
    ref = r.get("reference") or {}
    return {"value": ref.get("value"), "unit": ref.get("unit"), "cores": ref.get("cores"), "host_cores": ref.get("host_cores"), "kind": "reference",
            "port_on_the_same_host": (r.get("port") or {}).get("value"), "max_abs_reference_minus_port": r.get("max_abs_reference_minus_port"),
            "file": REFERENCE_RECORD, "note": "the reference module itself, timed where /root/reference is mounted (another host than this run's)"}


LINE_LIMIT = 6000       # bytes: the driver parses the LAST stdout line; round 3's 26 KB line did not parse


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 1] + "…"


def _b(bound):
    return "hbm" if str(bound).startswith("hbm") else "mfma"


def _contract_line(full):
    """The keys the driver's contract names, and nothing that can fail on a partial record."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "data")
    line = {k: full[k] for k in keep if k in full}
    line["dtype"] = _short(str(full.get("dtype", "")), 120)
    cfg = full.get("config") or {}
    # (graph_probe — the host-enqueue probe behind --graphs auto — stays in the detail file: host_enqueue_ms_per_step is in the line)
    line["config"] = {k: (_short(v, 200) if isinstance(v, str) else v) for k, v in cfg.items() if k != "graph_probe"} if isinstance(cfg, dict) else {"workload": _short(str(cfg), 200)}
    return line


def _optional_blocks(full, line):
    """roofline / rooflines / cpu_baseline / parity / fp32_exact / c4 / sharded blocks of the compact line (may raise on an unexpected record)."""
    for k in ("effective_tflops", "host_enqueue_ms_per_step", "host_ms_per_step_in_timed_region_incl_queue_backpressure", "output_checksum", "output_sane", "strong_error"):
        if k in full:
            line[k] = full[k]
    r = full.get("roofline")
    if r:
        rr = {k: r.get(k) for k in ("bound", "kind", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_tflops", "mfma_passes_per_product",
                                    "launches", "avg_launch_us", "kernel_ms_per_step", "share_of_step") if k in r}
        rr["bound"] = _b(rr.get("bound", ""))
        rr["kernel"] = _short(r.get("kernel", ""), 90)
        hb = r.get("hbm") or {}
        if "bytes_per_launch" in hb:
            rr["algorithmic_bytes_per_launch"] = hb["bytes_per_launch"]
        if rr.get("traffic") and hb.get("bytes_per_launch"):
            rr["traffic_over_algorithmic"] = round(rr["traffic"] / hb["bytes_per_launch"], 3)
        ts = r.get("traffic_source") or {}
        if ts:
            rr["traffic_source"] = {"file": ts.get("file"), "git_head": ts.get("git_head"), "tree": ts.get("tree")}
            if ts.get("warning"):
                rr["traffic_source"]["stale"] = True
        su = r.get("sustained") or {}
        if su:
            rr["sustained"] = {k: su.get(k) for k in ("peak", "clock_ghz", "frac") if k in su}
        line["roofline"] = rr
    if full.get("rooflines"):
        line["rooflines"] = [dict({"kind": x["kind"], "bound": _b(x["bound"]), "frac": x["frac"], "achieved": x["achieved"], "unit": x["unit"],
                                   "ms": x["kernel_ms_per_step"]}, **({"traffic_x": x["traffic_over_algorithmic"]} if x.get("traffic_over_algorithmic") else {}))
                             for x in full["rooflines"]]      # traffic_x: rocprofv3 counter bytes / algorithmic bytes per launch (HBM kinds)
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: (_short(v, 200) if k == "sample" else v) for k, v in cb.items() if k in ("value", "unit", "cores", "host_cores", "kind", "sample", "port_equals_reference")}
        if cb.get("reference_record"):
            line["cpu_baseline"]["reference_record"] = {k: cb["reference_record"].get(k) for k in ("value", "cores", "kind", "port_on_the_same_host", "file")}
    pv = full.get("parity_vs_cpu_oracle")
    if pv:
        line["parity_vs_cpu_oracle"] = {k: pv[k] for k in ("window", "frames", "max_abs_diff", "ref_max_abs", "psnr_db_uint8") if k in pv}
    for name in ("fp32_exact", "f16"):
        f = full.get(name)
        if f:
            line[name] = {"value": f["value"], "ms_per_step": f["ms_per_step"], "roofline_frac": (f.get("roofline") or {}).get("frac"),
                          "roofline_kind": (f.get("roofline") or {}).get("kind"), "composite_vs_headline": f.get("composite_vs_headline")}
    c4 = full.get("c4")
    if c4:
        if "error" in c4:
            line["c4"] = {"error": _short(c4["error"], 300)}
        else:
            st = {}
            for k, v in c4.get("stages", {}).items():
                e = {kk: vv for kk, vv in v.items() if kk.startswith("ms_per_")}
                rl = v.get("roofline") or {}
                if rl:
                    e["bound"] = _b(rl.get("bound", ""))
                    # MFMA stages: `frac_effective` = reference FLOP count (incl. encoder passes the pipeline no longer executes) x MFMA passes
                    # over the stage time; `frac_hardware` = the executed conv launches' event-timed rate (c4.rooflines).  HBM stages: `frac`.
                    if rl.get("bound") == "mfma":
                        e["frac_effective"] = rl.get("frac")
                        if "frac_hardware" in rl:
                            e["frac_hardware"] = rl["frac_hardware"]
                    else:
                        e["frac"] = rl.get("frac")
                st[k] = e
            cc = {"stages": st, "checksum": c4.get("checksum"), "pipeline_frames_per_s": (c4.get("pipeline_frames_per_s") or {}).get("value"),
                  "pipeline_ms_per_clip": (c4.get("pipeline_frames_per_s") or {}).get("ms_per_clip"),
                  "rooflines": [dict({"kind": _short(x["kind"], 40), "bound": x["bound"], "frac": x["frac"], "achieved": x["achieved"], "unit": x["unit"]},
                                     **({"traffic_x": x["traffic_over_algorithmic"]} if x.get("traffic_over_algorithmic") else {}))
                                for x in c4.get("rooflines", [])]}
            cpu = c4.get("cpu_baseline") or {}
            if cpu:
                # per stage: the CPU time of its unit in ms (absolute: ms per flow / pair / map / frame, or per clip where the sample was
                # extrapolated) next to the GPU speed-up over it
                unit_ms = lambda v: next((v[k] for k in ("ms_per_flow", "ms_per_pair", "ms_per_map", "ms_per_frame", "ms_per_clip_extrapolated") if k in v), None)
                # (the GPU speed-ups are the ratio of this to the stage's ms per unit above: in the detail file only, the line is at its size limit)
                cc["cpu_baseline"] = {"cores": cpu.get("cores"), "kind": cpu.get("kind"),
                                      "cpu_ms_per_unit": {k: unit_ms(v) for k, v in cpu.items() if isinstance(v, dict)}}
            c2 = c4.get("c2_spatial_mhsa")
            if c2:
                cc["c2_spatial_mhsa"] = {k: c2[k] for k in ("ms_per_module", "mfma_frac_module_wall", "mfma_frac_attention_kernel",
                                                            "hbm_frac_attention_kernel", "error", "streams_in_graph_replay", "mfma_frac_module_wall_graph_replay") if k in c2}
                if "fp32_exact" in c2:
                    cc["c2_spatial_mhsa"]["fp32_exact_mfma_frac_module_wall"] = c2["fp32_exact"].get("mfma_frac_module_wall")
                    cc["c2_spatial_mhsa"]["fp32_exact_mfma_frac_module_wall_graph_replay"] = c2["fp32_exact"].get("mfma_frac_module_wall_graph_replay")
                if "at_step_size_t136" in c2:
                    cc["c2_spatial_mhsa"]["t136_mfma_frac_module_wall"] = c2["at_step_size_t136"].get("mfma_frac_module_wall")
                    cc["c2_spatial_mhsa"]["t136_hbm_frac_attention_kernel"] = c2["at_step_size_t136"].get("hbm_frac_attention_kernel")
            line["c4"] = cc
    for k in ("phases_ms", "weak_scaling_clip_per_rank", "strong_scaling_same_clip", "strong_scaling_ideal", "pipeline_sharded", "feature_exchange",
              "strong_scaling_modes", "host_launch_us_probe", "rccl_preflight", "step_profile"):
        if k in full:
            v = full[k]
            if k == "strong_scaling_ideal":
                v = {kk: vv for kk, vv in v.items() if kk != "note"}
            line[k] = v


def compact_line(full, detail_path=None):
    """The ONE JSON line of the bench contract, derived from the full record `full` (which goes to `detail_path`): contract keys,
    `roofline` of the dominant kernel, compact `rooflines` (kind / bound / frac / ms), `cpu_baseline`, `parity_vs_cpu_oracle`,
    `fp32_exact` {value, ms_per_step, roofline_frac}, `c4` {per stage: ms + hardware / effective fractions, pipeline frames/s}.
    Always < LINE_LIMIT bytes (tests/test_bench_line.py): optional blocks are dropped from the tail if a future field overflows it, and a record
    the optional blocks cannot digest (a missing key, an unexpected shape) still yields the contract keys — a final line is ALWAYS printed."""
    line = _contract_line(full)
    try:
        _optional_blocks(full, line)
    except Exception as e:  # noqa: BLE001 — whatever a partial record throws, the contract keys go out
        line = _contract_line(full)
        line["compact_error"] = _short(f"{type(e).__name__}: {e}", 200)
        core = {"roofline": ("bound", "kind", "achieved", "peak", "unit", "frac", "traffic"), "cpu_baseline": ("value", "unit", "cores", "kind", "sample")}
        for k, keys in core.items():                    # the two objects the contract adds: their scalar core
            v = full.get(k)
            if isinstance(v, dict):
                line[k] = {kk: (_short(v[kk], 200) if isinstance(v[kk], str) else v[kk]) for kk in keys if kk in v and not isinstance(v[kk], (dict, list))}
    if detail_path:
        line["detail"] = detail_path
    # safety net: never exceed the limit — drop optional blocks from the least important end, then fall back to the contract keys alone
    for k in ("phases_ms", "f16", "rooflines", "strong_scaling_ideal", "c4", "fp32_exact", "parity_vs_cpu_oracle", "pipeline_sharded", "strong_scaling_same_clip",
              "weak_scaling_clip_per_rank", "feature_exchange", "strong_scaling_modes", "rccl_preflight"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        if k in line:
            line.pop(k)
            line.setdefault("dropped", []).append(k)
    if len(json.dumps(line)) >= LINE_LIMIT:
        dropped = sorted(set(line) - set(_contract_line(full)))
        line = _contract_line(full)
        line["config"] = {k: v for k, v in line["config"].items() if not isinstance(v, (str, dict, list))}
        line["dropped"] = dropped
    return line


def graphs_enabled(mode, world, host_s=None, step_s=None):
    """The --graphs policy.  'on' / 'off' as given; 'auto': N > 1 always (a rank's share of the FGT step is 13-20 ms of kernels behind ~600 launches:
    enqueueing them costs a slow host more than that), N = 1 when the probe step says the step is within 2x of launch-bound (host enqueue time >=
    half of the step's wall time: the driver's round-4 box, 67 of 102 ms; the builder's boxes need 9)."""
    if mode in ("on", "off"):
        return mode == "on"
    if world > 1:
        return True
    return host_s is not None and step_s is not None and step_s > 0 and host_s >= 0.5 * step_s


def emit(full):
    """Write the full record to gpurun_out/bench_detail.json (scratch dir of a GPU visit; created if missing) and print the compact line LAST."""
    rel = os.path.join("gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, rel), "w") as f:
            json.dump(full, f, indent=1)
    except OSError:
        rel = None
    if os.environ.get("FGT_BENCH_DETAIL_STDOUT") == "1":
        print("BENCH_DETAIL " + json.dumps(full), flush=True)       # a line BEFORE the last one
    line = compact_line(full, rel)
    print(json.dumps(line), flush=True)
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=80)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=432)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the per-launch HIP-event timing of the MFMA kernels")
    ap.add_argument("--precision", default="bf16x3", choices=["fp32", "bf16x3", "f16"],
                    help="arithmetic of the headline: exact fp32 MFMA; fp32 operands split into hi/lo bf16 with 3 bf16 MFMAs per "
                         "product and fp32 accumulation (FGT max |diff| vs reference 1.6e-6, bar 1e-3); or f16: GEMM / attention operands "
                         "rounded once to fp16 by their producer, one fp16 MFMA per product, fp32 accumulation (~1e-4, bar 1e-3)")
    ap.add_argument("--no-fp32-exact", action="store_true", help="N = 1: do not also time the exact-fp32 mode (the `fp32_exact` object)")
    ap.add_argument("--f16", action="store_true", help="N = 1: also time the f16 mode (the `f16` object; lower precision than the reference, never the headline)")
    ap.add_argument("--no-f16", action="store_true", help="(default since round 4; kept so that older command lines still parse)")
    ap.add_argument("--no-c4", action="store_true", help="N = 1: skip the `c4` object (BASELINE config C4 + tool stages: RAFT, diffusion fill, LAFC, "
                                                         "gradient propagation, Poisson blend, pipeline frames/s)")
    ap.add_argument("--no-cache", action="store_true", help="recompute the per-frame encoders in every window like the reference")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N>1 headline: strong = one clip sharded by frames/windows over the ranks (default); weak = one clip per rank")
    ap.add_argument("--window-batch", type=int, default=8, help="equal-length windows per transformer+decoder forward (bit-identical results)")
    ap.add_argument("--encode-chunk", type=int, default=40, help="frames per call of the per-frame stages (conv encoders + soft split; 20 -> 40: +0.5 %%, bit-identical)")
    ap.add_argument("--graphs", nargs="?", const="on", default="auto", choices=["auto", "on", "off"],
                    help="replay each window group's launch sequence as a hipGraph (the roofline blocks are then measured on one extra eager step "
                         "after the timed region).  auto (default): on for N > 1 — a rank's share of the step is then shorter than the time the host "
                         "needs to enqueue it — and at N = 1 when a probe step shows the host needing >= half of the step to enqueue it")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="no GPU: time the reference module and the oracle port on window 0, write "
                                                                     "profiles/cpu_baseline_reference.json (run where /root/reference is mounted)")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        return cpu_baseline_only(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # FGT_BENCH_SHARE_GPU=1 + FGT_BENCH_BACKEND=gloo: rehearse the N-rank sharded path on a single-GPU box (all ranks on
    # cuda:0, collectives staged through the host).  The driver's multi-GPU runs use the defaults: one GPU per rank, RCCL.
    share = os.environ.get("FGT_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("FGT_BENCH_BACKEND", "nccl")
    local = 0 if share else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.set_grad_enabled(False)

    from fgt_amd import ops
    from fgt_amd.fgt_model import DEFAULT_CONFIG, Model
    from fgt_amd.scheduler import ClipRunner, ideal_speedup
    from fgt_amd.synth import synth_clip, synth_state_dict

    cfg = dict(DEFAULT_CONFIG, input_resolution=(240, 432))
    model = Model(cfg).eval()
    sd = synth_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        tt = torch.tensor([x], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return tt.item()

    def make_runner(sharded, seed, exchange=None):
        fr, fl, ms = synth_clip(args.frames, args.height, args.width, seed=seed, device=dev)
        r = ClipRunner(model, fr, fl, ms, rank=rank if sharded else 0, world=world if sharded else 1, cache_features=not args.no_cache,
                       use_graphs=graphs_enabled(args.graphs, world), window_batch=args.window_batch, encode_chunk=args.encode_chunk, exchange=exchange)
        return r, (fr, fl, ms)

    def launch_probe():
        """Host cost of ONE kernel launch on this box (us): 2000 back-to-back launches of a 16-byte kernel, host time only.  Printed next to
        host_enqueue_ms_per_step so that a slow-host box explains itself (round 4: 110 us per launch on the driver's box, 15 on the builder's)."""
        t = torch.zeros(1, 4, device=dev)
        for _ in range(50):
            ops.axpby(t, 1.0, out=t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            ops.axpby(t, 1.0, out=t)
        us = 1e6 * (time.perf_counter() - t0) / 2000
        torch.cuda.synchronize()
        return round(us, 2)

    def timed(runner, prec, prof):
        """prepare pass + W warm-up + K timed steps in arithmetic `prec`; returns (seconds max over ranks, host enqueue s, comp, prof dict)."""
        ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = prec
        runner.run()                  # untimed preparation pass: weight packing + per-shape tile autotuning (setup, not a step)
        barrier()
        for _ in range(args.warmup):
            runner.run()
        barrier()
        # probe step: the host's OWN cost of enqueueing one step — against an idle queue, no event records — next to the step's wall time.
        # (The timed region below enqueues K steps back to back: once the HIP queue holds its fill of launches the host BLOCKS in the launch
        # calls until the GPU drains it, so host time per step there grows with K — 9 ms at K = 5, 31 ms at K = 10, 67 ms at K = 20 on the same
        # kind of box (BENCH_r04) — and says nothing about how close the step is to launch-bound.  This probe does.)
        t0 = time.perf_counter()
        runner.run()
        h = time.perf_counter() - t0
        torch.cuda.synchronize()
        runner.graph_probe = {"host_enqueue_ms": round(1e3 * h, 3), "step_ms": round(1e3 * (time.perf_counter() - t0), 3), "graphs": bool(runner.use_graphs)}
        if args.graphs == "auto" and world == 1 and not runner.use_graphs and runner.cache_features and graphs_enabled("auto", 1, h, time.perf_counter() - t0):
            runner.use_graphs = True
            runner.run()                      # capture pass (untimed)
            barrier()
        if prof:
            ops.prof_collect("all")
            ops.prof_enable(True, kinds=list(KERNELS))      # the timed steps carry event pairs on the MFMA launches only (~250 per step)
        # the shader clock under full matrix load right before and right after the timed region, and an event at every step boundary: a step that is
        # power-limited slows down over a long timed region (the driver's 20 steps against the 3-5 of a quick A/B) — the line shows it instead of hiding it
        clk = {}
        try:
            clk["before_timed_region"] = round(ops.mfma_probe(f32=(prec == "fp32"), device=dev)[1], 3)
        except Exception:  # noqa: BLE001
            pass
        barrier()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            comp = runner.run()
            marks[i + 1].record()
        host_dt = time.perf_counter() - t0      # time the host needed to enqueue everything (launch-bound if close to dt)
        barrier()
        dt = time.perf_counter() - t0
        try:
            clk["after_timed_region"] = round(ops.mfma_probe(f32=(prec == "fp32"), device=dev)[1], 3)
        except Exception:  # noqa: BLE001
            pass
        per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        k3 = max(1, min(3, args.steps // 2))
        runner.step_profile = {"first_steps_ms": round(sum(per_step[:k3]) / k3, 3), "last_steps_ms": round(sum(per_step[-k3:]) / k3, 3),
                               "min_ms": round(min(per_step), 3), "max_ms": round(max(per_step), 3), "averaged_over": k3, "mfma_probe_clock_ghz": clk}
        kinds, scale = {}, 1
        if prof:
            if runner.use_graphs:     # graph replays bypass the per-launch events: measure the same kernels on one eager step
                ops.prof_collect("all")
                runner.use_graphs = False
                runner.run()
                torch.cuda.synchronize()
                runner.use_graphs = True
                scale = args.steps
            ops.prof_enable(False)
            for k in KERNELS:
                ms, fl, n, by = ops.prof_collect(k)
                kinds[k] = (ms * scale, fl * scale, n * scale, by * scale)
            # the HBM-bound kernels on ONE extra eager step outside the timed region (an event pair around every small launch would
            # slow the timed steps by a few per cent); scaled to the timed step count like the graph case above
            ops.prof_enable(True, kinds=list(HBM_KERNELS))
            ug, runner.use_graphs = runner.use_graphs, False
            runner.run()
            torch.cuda.synchronize()
            runner.use_graphs = ug
            ops.prof_enable(False)
            for k in HBM_KERNELS:
                ms, fl, n, by = ops.prof_collect(k)
                kinds[k] = (ms * args.steps, fl * args.steps, n * args.steps, by * args.steps)
        # per-phase times of ONE extra pass on every rank (encode / gather wait / windows / exchange / blend: events on the main stream),
        # so that a multi-GPU curve can be read phase by phase
        runner.timing = True
        runner.run()
        torch.cuda.synchronize()
        runner.timing = False
        ph = runner.phase_ms()
        ph["host_enqueue_per_step"] = round(1e3 * host_dt / max(args.steps, 1), 3)        # this rank's host time to enqueue one timed step
        if world > 1:
            allph = [None] * world
            dist.all_gather_object(allph, ph)
            ph = allph
        runner.last_phases = ph
        return max_over_ranks(dt), host_dt, comp, kinds

    sustained_cache = {}

    def sustained_peak(prec):
        """The matrix-core rate THIS chip sustains under load (fgt_mfma_probe: back-to-back MFMAs on random operands, clock measured in
        the kernel).  The nominal peaks assume 2.4 GHz; the chip clocks to its power budget, so this is the reachable roof."""
        if prec not in sustained_cache:
            try:
                tf, ghz = ops.mfma_probe(f32=(prec == "fp32"), device=dev)      # (f16: the bf16 probe — same instruction shape and rate)
                sustained_cache[prec] = {"peak": round(tf, 1), "clock_ghz": round(ghz, 3),
                                         "how": "fgt_mfma_probe: 256 workgroups x 8 wavefronts of independent "
                                                + ("v_mfma_f32_32x32x2_f32" if prec == "fp32" else "v_mfma_f32_32x32x16_bf16")
                                                + " on random operands in registers; clock = d(s_memtime)/d(s_memrealtime)"}
            except Exception as e:  # noqa: BLE001 - a diagnostic must not take the bench line down
                sustained_cache[prec] = {"peak": None, "error": str(e)[:200]}
        return sustained_cache[prec]

    def rooflines(kinds, prec, dt):
        passes = 3 if prec == "bf16x3" else 1        # MFMA flops issued per algorithmic flop (f16 mode: the few GEMMs that still take
        #                                              fp32 inputs — 4-channel input convs, re-weighting Linear, vec2patch — issue 3; counted as 1)
        peak = PEAK_FP32_MFMA_TFLOPS if prec == "fp32" else PEAK_BF16_MFMA_TFLOPS
        traffic = kernel_traffic(prec)
        sus = sustained_peak("fp32" if prec == "fp32" else "bf16x3") if kinds else None
        out = []
        for k, (ms, fl, n, by) in kinds.items():
            if ms <= 0 or n == 0:
                continue
            if k in HBM_KERNELS:
                gbs = by / (ms * 1e-3) / 1e9
                out.append({"bound": "hbm", "kernel": HBM_KERNELS[k], "kind": k, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                            "frac": round(gbs / PEAK_HBM_GBPS, 4), "algorithmic_bytes_per_launch": round(by / n), "launches": n,
                            "avg_launch_us": round(1e3 * ms / n, 2), "kernel_ms_per_step": round(ms / args.steps, 3), "share_of_step": round(ms / (1e3 * dt), 3)})
                # counter bytes per launch of the same kind from the rocprofv3 PMC passes over this command (profiles/kernel_traffic.json)
                tb = traffic.get(k, {}).get("hbm_bytes_per_launch")
                if tb:
                    out[-1].update(traffic=tb, traffic_over_algorithmic=round(tb / max(by / n, 1.0), 3),
                                   counter_GBps=round(tb * n / (ms * 1e-3) / 1e9, 1), traffic_source=traffic.get("_source"))
                continue
            ach = passes * fl / (ms * 1e-3) / 1e12
            gbs = by / (ms * 1e-3) / 1e9
            # the same launches on the HBM roofline: unique bytes (every input / weight / output byte once) over the same time
            hbm = {"algorithmic_GBps": round(gbs, 1), "peak_GBps": PEAK_HBM_GBPS, "frac": round(gbs / PEAK_HBM_GBPS, 4), "bytes_per_launch": round(by / n)}
            out.append({"bound": "mfma" if ach / peak >= gbs / PEAK_HBM_GBPS else "hbm (closer to the HBM roof than to the MFMA roof: see `hbm`)",
                        "kernel": KERNELS[k], "kind": k, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(ach / peak, 4), "hbm": hbm, "traffic": traffic.get(k, {}).get("hbm_bytes_per_launch"),
                        "traffic_source": traffic.get("_source"),
                        "algorithmic_tflops": round(fl / (ms * 1e-3) / 1e12, 2), "mfma_passes_per_product": passes,
                        "launches": n, "avg_launch_us": round(1e3 * ms / n, 2), "kernel_ms_per_step": round(ms / args.steps, 3),
                        "share_of_step": round(ms / (1e3 * dt), 3)})
            if sus:
                out[-1]["sustained"] = dict(sus, frac=round(ach / sus["peak"], 4)) if sus.get("peak") else sus
        out.sort(key=lambda r: -r["share_of_step"])
        return [r for r in out if r["kind"] in KERNELS] + [r for r in out if r["kind"] in HBM_KERNELS]

    def assemble(res, weak, strong_error=None):
        dt, runner = res["dt"], res["runner"]
        mult = world if (weak and world > 1) else 1
        fps = args.frames * args.steps / dt * mult
        sched = runner.sched
        clip_flops = sum(fgt_flops(len(a) + len(b)) for a, b in sched) if (args.height, args.width) == (240, 432) else None
        line = {
            "metric": f"inpainted frames/sec at {args.width}x{args.height}x{args.frames} clip (FGT stage: {len(sched)} sliding-window forwards + compose/blend)",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None,
            "dtype": DTYPES[prec],
            "data": "synthetic",
            "config": {"workload": f"full FGT forward, random (N(0,0.02)) weights, {args.width}x{args.height}x{args.frames} clip, "
                                   f"reference window schedule (neighbor_stride 5, step 10; sum t = {sum(len(a) + len(b) for a, b in sched)})",
                       "windows": len(sched),
                       "sharding": ("single GPU" if world == 1 else f"one clip per rank x {world} ranks, no data-path collective" if weak else
                                    f"one clip: frames block-sharded for the per-frame stages (chunked all-to-all of the feature rows each rank's windows "
                                    f"reference), windows cost-balanced over {world} ranks with equal lengths co-located, uint8 all-gather of window outputs"),
                       "conv_precision": prec, "per_frame_feature_cache": bool(runner.cache_features), "hip_graphs": bool(runner.use_graphs),
                       "graphs_mode": args.graphs, "window_batch": runner.window_batch},
            # the host's own cost of enqueueing ONE step against an idle queue (probe step, see timed()); the K-step figure beside it includes the
            # time the host spends blocked on the full launch queue (back-pressure of a GPU-bound step), which grows with K
            "host_enqueue_ms_per_step": (getattr(runner, "graph_probe", None) or {}).get("host_enqueue_ms"),
            "host_ms_per_step_in_timed_region_incl_queue_backpressure": round(1e3 * res["host_dt"] / args.steps, 3),
            "host_launch_us_probe": host_launch_us,
        }
        if getattr(runner, "graph_probe", None):
            line["config"]["graph_probe"] = runner.graph_probe
        if getattr(runner, "step_profile", None):
            line["step_profile"] = runner.step_profile
        if clip_flops:
            line["effective_tflops"] = round(clip_flops * args.steps * mult / dt / 1e12, 2)
        rl = rooflines(res["kinds"], prec, dt)
        if rl:
            line["roofline"] = dict(rl[0], note="kernel with the largest share of the step (rank 0's launches of the timed region)")
            line["rooflines"] = rl
        ph = getattr(runner, "last_phases", None)
        if ph:
            line["phases_ms"] = ph if isinstance(ph, dict) else {f"rank{i}": p_ for i, p_ in enumerate(ph)}
            if not isinstance(ph, dict):
                line["feature_rows_per_rank"] = {"rows_held": getattr(runner, "rows", None), "frames": args.frames,
                                                 "note": "a rank receives only the frames its windows reference (needed-rows all-to-all)"}
        if strong_error:
            line["strong_error"] = strong_error
        c = res["comp"].float()
        line["output_checksum"] = round(float(c.double().mean()), 6)      # rank 0's clip (seed 1234): identical for every N and both scalings
        line["output_sane"] = bool(torch.isfinite(c).all() and c.min().item() >= 0 and c.max().item() <= 255)
        return line

    prec = args.precision
    n_sched = None
    out = None
    weak_res = strong_res = None
    host_launch_us = None if args.no_prof else launch_probe()        # (--no-prof: the rocprofv3 --pmc passes: no probe launches in their traces)

    # ---------------------------------------------------------------- N = 1 (and the weak, clip-per-rank variant at N > 1)
    runner, clip = make_runner(False, 1234 + (rank if world > 1 else 0))
    n_sched = runner.sched
    dt, host_dt, comp, kinds = timed(runner, prec, prof=not args.no_prof)
    weak_res = dict(dt=dt, host_dt=host_dt, comp=comp, kinds=kinds, runner=runner, clip=clip)

    # ---------------------------------------------------------------- N > 1: ONE clip sharded over the ranks, once per feature exchange
    # First with the plain per-chunk all-gather (equal-sized contributions: the most ordinary collective), then with the needed-rows
    # all-to-all (uneven / zero-length splits: never run on RCCL hardware by the builder).  Each under its own watchdog: if the second one
    # hangs, the line goes out with the first one's result as the headline instead of the clip-per-rank number.
    strong_err = None
    strong_modes = {}
    preflight = None
    if world > 1:
        # ---- collective pre-flight (fgt_amd/preflight.py): the sharded clip's three collectives at its sizes, 3 reps each, every call under a 10-s
        # watchdog — a first run on RCCL that hangs or fails still yields a line that says WHICH collective it was
        from fgt_amd.preflight import collective_preflight
        per = -(-args.frames // world)

        def preflight_hang(name, partial):
            if rank == 0:
                partial["failed"] = name
                partial["checks"][name] = {"error": "no completion within the watchdog (collective hung)"}
                emit(dict(assemble(weak_res, weak=True, strong_error=f"collective pre-flight: {name} hung"), rccl_preflight=partial))
            os._exit(0)

        preflight = collective_preflight(dev, rank, world, frame_floats=(args.height // 4) * (args.width // 4) * 128, frames_per_rank=min(per, args.encode_chunk),
                                         u8_bytes_per_rank=2 * 6 * 3 * args.height * args.width, reps=3, timeout_s=10.0, on_hang=preflight_hang)
        if preflight["failed"] and not os.environ.get("FGT_EXCHANGE"):
            # the all-to-all failed but the all-gathers work: only the all-gather exchange is attempted (the conservative path); an all-gather
            # failure leaves nothing to shard with — the clip-per-rank line goes out with the reason
            if preflight["failed"].startswith("all_to_all"):
                os.environ["FGT_EXCHANGE"] = "allgather"
            else:
                strong_err = f"collective pre-flight failed: {preflight['failed']}: {preflight['checks'][preflight['failed']].get('error')}"
    if world > 1 and not (preflight and preflight["failed"] and not preflight["failed"].startswith("all_to_all")):
        modes = [os.environ["FGT_EXCHANGE"].lower()] if os.environ.get("FGT_EXCHANGE") else ["allgather", "a2a"]
        for mode in modes:
            def give_up(mode=mode):
                if rank == 0:
                    err = f"sharded section ({mode}) timed out after 300 s"
                    if strong_res is not None:
                        o = assemble(strong_res, weak=False, strong_error=err)
                        o["feature_exchange"] = strong_res["exchange"]
                        o["strong_scaling_modes"] = strong_modes
                        emit(o)
                    else:
                        emit(assemble(weak_res, weak=True, strong_error=err))
                os._exit(0)

            dog = threading.Timer(300.0, give_up)
            dog.daemon = True
            try:
                r2, clip2 = make_runner(True, 1234, exchange=mode)
                dog.start()
                dt2, host2, comp2, kinds2 = timed(r2, prec, prof=not args.no_prof)
                res2 = dict(dt=dt2, host_dt=host2, comp=comp2, kinds=kinds2, runner=r2, clip=clip2, exchange=mode)
                strong_modes[mode] = {"value": round(args.frames * args.steps / dt2, 3), "ms_per_step": round(1e3 * dt2 / args.steps, 3),
                                      "host_enqueue_ms_per_step": (getattr(r2, "graph_probe", None) or {}).get("host_enqueue_ms"), "hip_graphs": bool(r2.use_graphs),
                                      "output_checksum": round(float(comp2.double().mean()), 6)}
                if strong_res is None or dt2 < strong_res["dt"]:
                    strong_res = res2
            except Exception as e:      # the weak number (or the other mode) stands on its own; report instead of losing the line
                strong_err = f"{mode}: {type(e).__name__}: {e}"[:300]
                strong_modes[mode] = {"error": strong_err}
            dog.cancel()

    # ---------------------------------------------------------------- N > 1: the flow stages of the covered chain, sharded (watchdog-guarded too)
    pipe_sharded = None
    if world > 1 and not args.no_c4:
        def pipe_give_up():
            if rank == 0:
                hw = args.scaling == "weak" or strong_res is None
                emit(dict(assemble(weak_res if hw else strong_res, weak=hw, strong_error=strong_err), pipeline_sharded={"error": "timed out after 300 s"}))
            os._exit(0)

        dog2 = threading.Timer(300.0, pipe_give_up)
        dog2.daemon = True
        dog2.start()
        try:
            import bench_stages
            pipe_sharded = bench_stages.run_sharded(dev, prec, rank, world, frames=args.frames, H=args.height, W=args.width,
                                                    fgt_ms=(1e3 * strong_res["dt"] / args.steps) if strong_res is not None else None, backend=backend)
        except Exception as e:  # noqa: BLE001 - the side object must not take the headline down
            pipe_sharded = {"error": f"{type(e).__name__}: {e}"[:300]}
        dog2.cancel()

    if rank == 0:
        headline_weak = world == 1 or args.scaling == "weak" or strong_res is None
        out = assemble(weak_res if headline_weak else strong_res, weak=headline_weak, strong_error=strong_err)
        if pipe_sharded is not None:
            out["pipeline_sharded"] = pipe_sharded
        if world > 1:
            out["feature_exchange"] = getattr(strong_res["runner"], "exchange", None) if strong_res is not None else None
            out["strong_scaling_modes"] = strong_modes        # the sharded clip per feature exchange (the headline is the faster one)
            out["rccl_preflight"] = preflight
            if not headline_weak:
                # (the headline of an N > 1 run is the FASTER of the two exchanges — say which one in the contract keys)
                out["config"]["feature_exchange"] = strong_res.get("exchange")
                out["config"]["headline_is_best_of_exchanges"] = sorted(strong_modes)
            bound = ideal_speedup(n_sched, world)
            per = -(-args.frames // world)
            side = lambda res, weak: {"value": round(args.frames * args.steps / res["dt"] * (world if weak else 1), 3), "unit": "frames/s",
                                      "ms_per_step": round(1e3 * res["dt"] / args.steps, 3), "scaling": "weak" if weak else "strong",
                                      "output_checksum": round(float(res["comp"].double().mean()), 6)}
            if strong_res is not None and not headline_weak:
                out["weak_scaling_clip_per_rank"] = side(weak_res, True)
            elif strong_res is not None:
                out["strong_scaling_same_clip"] = side(strong_res, False)
            out["strong_scaling_ideal"] = {"window_phase_speedup_bound": round(bound, 3), "frame_phase_speedup_bound": round(args.frames / per, 3),
                                           "note": f"{len(n_sched)} windows over {world} ranks: total window cost / most loaded rank (scheduler.assign_windows); "
                                                   f"per-frame stages {args.frames} frames / {per} per rank"}

    # ---------------------------------------------------------------- N = 1 extras: exact-fp32 mode, CPU baseline + parity
    if world == 1 and rank == 0:
        runner = weak_res["runner"]
        frames, flows, masks = weak_res["clip"]
        parity_of = None
        if not args.no_cpu_baseline:      # CPU baseline: rank 0 at N = 1 only (headline precision still selected)
            ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = prec
            out["parity_vs_cpu_oracle"], parity_of, out["cpu_baseline"] = cpu_baseline(cfg, sd, frames, flows, masks, runner.sched, model)
            if out["cpu_baseline"]["kind"] == "port":
                out["cpu_baseline"]["port_equals_reference"] = "tests/test_oracle_pinned.py"
                rr = reference_record()
                if rr:
                    out["cpu_baseline"]["reference_record"] = rr
        if prec != "f16" and args.f16 and not args.no_f16:
            # the third arithmetic mode in the same invocation: operands rounded once to fp16 by their producer, one MFMA per product
            dt16, host16, comp16, kinds16 = timed(runner, "f16", prof=not args.no_prof)
            rl = rooflines(kinds16, "f16", dt16)
            d8 = (comp16.float() - weak_res["comp"].float()).abs()
            out["f16"] = {"value": round(args.frames * args.steps / dt16, 3), "unit": "frames/s", "ms_per_step": round(1e3 * dt16 / args.steps, 3),
                          "dtype": DTYPES["f16"], "steps": args.steps, "warmup": args.warmup, "roofline": rl[0] if rl else None, "rooflines": rl,
                          "effective_tflops": round(sum(fgt_flops(len(a) + len(b)) for a, b in runner.sched) * args.steps / dt16 / 1e12, 2)
                          if (args.height, args.width) == (240, 432) else None,
                          "parity_vs_cpu_oracle": parity_of(model, "f16 mode") if parity_of else None,
                          "composite_vs_headline": {"max_uint8_steps": float(d8.max()), "differing_values": float((d8 > 0).float().mean())},
                          "output_sane": bool(torch.isfinite(comp16.float()).all())}
            ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = prec
        if prec != "fp32" and not args.no_fp32_exact:
            dt32, host32, comp32, kinds32 = timed(runner, "fp32", prof=not args.no_prof)
            rl = rooflines(kinds32, "fp32", dt32)
            d8 = (comp32.float() - weak_res["comp"].float()).abs()
            out["fp32_exact"] = {"value": round(args.frames * args.steps / dt32, 3), "unit": "frames/s", "ms_per_step": round(1e3 * dt32 / args.steps, 3),
                                 "dtype": "f32 (v_mfma_f32_32x32x2_f32: bit-exact fmaf chains)", "steps": args.steps, "warmup": args.warmup,
                                 "roofline": rl[0] if rl else None, "rooflines": rl,
                                 "effective_tflops": round(sum(fgt_flops(len(a) + len(b)) for a, b in runner.sched) * args.steps / dt32 / 1e12, 2)
                                 if (args.height, args.width) == (240, 432) else None,
                                 "composite_vs_headline": {"max_uint8_steps": float(d8.max()), "differing_values": float((d8 > 0).float().mean())}}
            ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = prec
        if not args.no_c4:
            # BASELINE config C4 + the tool stages around the FGT stage on the same clip geometry, each with its roofline and a CPU baseline
            try:
                import bench_stages
                out["c4"] = bench_stages.run_stages(dev, prec, frames=args.frames, H=args.height, W=args.width, fgt_ms=out["ms_per_step"],
                                                    with_cpu=not args.no_cpu_baseline, fgt_model=model)
            except Exception as e:  # noqa: BLE001 - the side object must not take the headline down
                import traceback
                out["c4"] = {"error": f"{type(e).__name__}: {e}"[:400], "trace": traceback.format_exc()[-1200:]}
        if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
            ops.save_tuning(os.path.join(ROOT, "gpurun_out", "tuning.json"))

    if rank == 0:
        emit(out)
        assert out["output_sane"], "composited clip has NaN/inf or out-of-range values"
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
