"""MI355X parity of the flow side: LAFC, RAFT, image_warp / fbConsistencyCheck and the RAFT helper kernels, against
golden vectors from the reference and the CPU oracle."""
import argparse
import json
import os

import pytest
import torch
import torch.nn.functional as F

from fgt_amd import lafc_model, raft_model
from fgt_amd.synth import synth_state_dict
from oracle import lafc_oracle as LO
from oracle import raft_oracle as RO
from util import GOLDEN, load_golden, max_err, rel_err, report

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _sd(name, mode="kaiming"):
    keys = json.load(open(os.path.join(GOLDEN, name)))
    tmpl = {k: torch.empty(v, dtype=torch.long if k.endswith("num_batches_tracked") else torch.float32) for k, v in keys.items()}
    return synth_state_dict(tmpl, seed=0, mode=mode)


@pytest.mark.parametrize("ct", ["vanilla", "gated"])
def test_lafc_matches_reference_golden(ct, dev):
    m = lafc_model.Model(dict(lafc_model.DEFAULT_CONFIG, conv_type=ct)).eval()
    m.load_state_dict(_sd(f"lafc_{ct}_state_keys.json"), strict=True)
    m = m.to(dev)
    g = load_golden(f"lafc_{ct}_64x96.npz")
    flow, edge = m(g["flows"].to(dev), g["masks"].to(dev))
    e1, r1 = report(f"lafc {ct} flow", flow, g["flow"])
    e2, r2 = report(f"lafc {ct} edge", edge, g["edge"])
    assert r1 < 1e-4 and e2 < 1e-4           # fp32 bar 1e-3 absolute; observed ~1e-6


def test_lafc_240x432_matches_oracle(dev):
    """BASELINE config #4 shape: one LAFC call on 3 flows at 240x432."""
    sd = _sd("lafc_vanilla_state_keys.json")
    m = lafc_model.Model(dict(lafc_model.DEFAULT_CONFIG)).eval()
    m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(7)
    fl = torch.randn(1, 2, 3, 240, 432, generator=g)
    ms = (F.interpolate((torch.rand(3, 1, 30, 54, generator=g) > 0.6).float(), size=(240, 432))).view(1, 3, 1, 240, 432).permute(0, 2, 1, 3, 4).contiguous()
    ref = LO.lafc_forward(sd, lafc_model.DEFAULT_CONFIG, fl * (1 - ms), ms)
    out = m.to(dev)((fl * (1 - ms)).to(dev), ms.to(dev))
    assert report("lafc 240x432 flow", out[0], ref[0])[1] < 1e-4 and report("lafc 240x432 edge", out[1], ref[1])[0] < 1e-4


def test_raft_matches_reference_golden(dev):
    m = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
    m.load_state_dict(_sd("raft_state_keys.json"), strict=True)
    m = m.to(dev)
    g = load_golden("raft_128x160_it6.npz")
    lo, up = m(g["image1"].to(dev), g["image2"].to(dev), iters=6, test_mode=True)
    e1, r1 = report("raft flow_low", lo, g["flow_low"])
    e2, r2 = report("raft flow_up", up, g["flow_up"])
    assert r1 < 1e-3 and r2 < 1e-3           # 6 GRU iterations amplify fp32 round-off; flows are O(100) px with these weights


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_raft_context_term_hoisted_out_of_the_loop_is_exact(prec, dev, monkeypatch):
    """RAFT/update.py:45-58: hx = [h | inp | motion] and `inp` never changes over the refinement loop (raft.py:112-115): the GRU convs are split
    into a per-iteration part over [h | motion] and a context part over inp (+ bias) evaluated once per pair and added as a bias map.  Exact in
    real arithmetic (a sum split in two); in fp32 the two forms differ by summation order only — both are held to the reference golden, and to
    each other well inside it."""
    from fgt_amd import ops
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", prec)
    g = load_golden("raft_128x160_it6.npz")
    outs = {}
    for hoist in (True, False):
        m = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
        m.load_state_dict(_sd("raft_state_keys.json"), strict=True)
        m = m.to(dev)
        m.hoist_context = hoist
        outs[hoist] = m(g["image1"].to(dev), g["image2"].to(dev), iters=6, test_mode=True)
        assert report(f"raft flow_up hoist={hoist} {prec}", outs[hoist][1], g["flow_up"])[1] < 1e-3
    scale = g["flow_up"].abs().max().item()
    d = (outs[True][1] - outs[False][1]).abs().max().item()
    print(f"[parity] RAFT {prec}: context term hoisted vs in the loop: max |diff| {d:.3e} px (flows up to {scale:.1f} px)")
    assert d < 3e-4 * scale


def test_raft_240x432_two_iterations_matches_oracle(dev):
    sd = _sd("raft_state_keys.json")
    m = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
    m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(8)
    base = F.interpolate(torch.rand(1, 3, 32, 56, generator=g), size=(248, 440), mode="bilinear", align_corners=False) * 255
    i1, i2 = base[:, :, 4:244, 4:436].contiguous(), base[:, :, 3:243, 6:438].contiguous()
    ref = RO.raft_forward(sd, i1, i2, iters=2)
    lo, up = m.to(dev)(i1.to(dev), i2.to(dev), iters=2, test_mode=True)
    assert report("raft 240x432 low", lo, ref[0])[1] < 1e-3 and report("raft 240x432 up", up, ref[1])[1] < 1e-3


def test_warp_and_fb_consistency_match_reference_golden(dev):
    from fgt_amd import ops
    g = load_golden("warp_24x40.npz")
    nhwc = lambda x: x.permute(0, 2, 3, 1).contiguous().to(dev)
    out = ops.warp(nhwc(g["img"]), nhwc(g["flow"]))
    assert report("image_warp", out.permute(0, 3, 1, 2), g["warped"])[0] < 1e-5
    o1, o2 = ops.fb_consistency(nhwc(g["f1"]), nhwc(g["f2"]))
    mism = ((o1.cpu() != g["occ_fw"][:, 0]).float().mean() + (o2.cpu() != g["occ_bw"][:, 0]).float().mean()).item()
    print(f"[parity] fb_consistency mismatching pixels fraction: {mism:.2e}")
    assert mism < 2e-3                       # thresholded output: only pixels within round-off of the threshold may flip


@pytest.mark.parametrize("C,H,W", [(2, 24, 40), (2, 17, 33), (3, 24, 40), (4, 9, 14), (8, 16, 20), (6, 11, 13)])
def test_warp_every_vector_width_matches_oracle(C, H, W, dev):
    """fgt_warp picks its access width from C and the alignment (C = 2 with even W: two pixels per work item; C % 4 == 0: 16-byte accesses;
    C % 2 == 0: 8-byte; else scalar): every path against the oracle's image_warp (LAFC/models/utils/fbConsistencyCheck.py:8-26), incl.
    flows that leave the image (zeros padding) and a channel slice of a wider buffer."""
    from fgt_amd import ops
    g = torch.Generator().manual_seed(100 + C + W)
    B = 3
    img = torch.randn(B, C, H, W, generator=g)
    flow = torch.randn(B, 2, H, W, generator=g) * 6.0
    ref = RO.image_warp(img, flow)
    nhwc = lambda x: x.permute(0, 2, 3, 1).contiguous().to(dev)
    out = ops.warp(nhwc(img), nhwc(flow))
    assert report(f"image_warp C={C} {W}x{H}", out.permute(0, 3, 1, 2), ref)[0] < 1e-5
    wide = torch.zeros(B, H, W, C + 5, device=dev)
    wide[..., 1:1 + C] = nhwc(img)
    assert torch.equal(ops.warp(wide[..., 1:1 + C], nhwc(flow)), out)          # an unaligned channel slice: the scalar path, same values
    fbuf = torch.zeros(B * H * W * 2 + 1, device=dev)
    fbuf[1:] = nhwc(flow).reshape(-1)
    assert torch.equal(ops.warp(nhwc(img), fbuf[1:].view(B, H, W, 2)), out)    # a contiguous flow view at an odd float offset (4-byte aligned only)


def test_raft_helper_kernels(dev):
    from fgt_amd import ops
    g = torch.Generator().manual_seed(3)
    B, H1, W1 = 2, 16, 24
    vol = torch.randn(B * H1 * W1, H1, W1, generator=g)
    p1 = ops.avgpool2(vol.to(dev), B * H1 * W1, H1, W1)
    assert max_err(p1, F.avg_pool2d(vol[:, None], 2, stride=2)[:, 0]) < 1e-6
    pyr = [vol[:, None]]
    for _ in range(3):
        pyr.append(F.avg_pool2d(pyr[-1], 2, stride=2))
    coords = RO.coords_grid(B, H1, W1) + torch.randn(B, 2, H1, W1, generator=g) * 3
    ref = RO.corr_lookup(pyr, coords)
    dpyr = [p[:, 0].contiguous().to(dev) for p in pyr]
    out = torch.empty(B, H1, W1, 324, device=dev)
    ops.corr_lookup(dpyr, B, H1, W1, 4, coords.permute(0, 2, 3, 1).contiguous().to(dev), out)
    assert report("corr_lookup", out.permute(0, 3, 1, 2), ref)[0] < 1e-4
    # split output (RAFT's update block in bf16x3 mode): the same values as a bf16 hi / lo pair, zero channels up to a multiple of 32;
    # both forms from one launch; coordinates far outside the map and exactly on integers (taps outside the staged window / zero weights)
    cs = ops.Split.empty((B, H1, W1, 352), dev, h=False)
    out2 = torch.empty_like(out)
    ops.corr_lookup(dpyr, B, H1, W1, 4, coords.permute(0, 2, 3, 1).contiguous().to(dev), out2, out_s=cs)
    assert torch.equal(out2, out)
    assert torch.equal(cs.data[..., :324], ops.split(out, h=False).data) and float(cs.data[..., 324:].float().abs().max()) == 0.0
    c2 = (RO.coords_grid(B, H1, W1) + torch.randn(B, 2, H1, W1, generator=g).round() * 7).clone()
    c2[:, :, :2] = 1e4
    c2[:, :, 2:4] = -37.5
    ref2 = RO.corr_lookup(pyr, c2)
    ops.corr_lookup(dpyr, B, H1, W1, 4, c2.permute(0, 2, 3, 1).contiguous().to(dev), out2)
    assert report("corr_lookup (integer / far coordinates)", out2.permute(0, 3, 1, 2), ref2)[0] < 1e-4
    flow = torch.randn(B, 2, H1, W1, generator=g)
    mask = torch.randn(B, 576, H1, W1, generator=g)
    f4 = torch.zeros(B, H1, W1, 4)
    f4[..., :2] = flow.permute(0, 2, 3, 1)
    up = ops.convex_upsample(f4.to(dev), mask.permute(0, 2, 3, 1).contiguous().to(dev))
    assert report("convex_upsample", up, RO.upsample_flow(flow, mask))[0] < 1e-4
    x = torch.randn(2, 96, 20, 28, generator=g) * 2 + 0.5
    res = torch.randn(2, 96, 20, 28, generator=g)
    ref = F.relu(F.relu(F.instance_norm(x)) + res)
    out = ops.instnorm(x.permute(0, 2, 3, 1).contiguous().to(dev), act="relu", res=res.permute(0, 2, 3, 1).contiguous().to(dev), act2="relu")
    assert report("instnorm", out.permute(0, 3, 1, 2), ref)[0] < 1e-4


def test_flow_pipeline_batched_pairs_match_per_pair_calls_gpu(dev):
    from fgt_amd import flow_pipeline
    m = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
    m.load_state_dict(_sd("raft_state_keys.json"), strict=True)
    m = m.to(dev)
    g = load_golden("raft_128x160_it6.npz")
    frames = torch.cat([g["image1"], g["image2"], g["image1"].flip(-1), g["image2"].flip(-2)], 0).to(dev)
    fw, bw = flow_pipeline.compute_flows(m, frames, iters=6, batch=4, enc_batch=3)
    assert report("flow pipeline fwd[0] vs reference golden", fw[0:1], g["flow_up"])[1] < 1e-3
    for i in range(3):
        a = m(frames[i:i + 1], frames[i + 1:i + 2], iters=6, test_mode=True)[1]
        b = m(frames[i + 1:i + 2], frames[i:i + 1], iters=6, test_mode=True)[1]
        assert report(f"flow pipeline fwd[{i}] vs per-pair call", fw[i:i + 1], a)[1] < 1e-4
        assert report(f"flow pipeline bwd[{i}] vs per-pair call", bw[i:i + 1], b)[1] < 1e-4


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_raft_tool_setting_864x480_twenty_iterations_vs_oracle(prec, dev, monkeypatch):
    """The tool-faithful RAFT call: frames resized to 2x (tool/video_inpainting.py:263 via :447-450 -> 864x480 for a 432x240 clip),
    iters = 20, test_mode.  Twenty GRU iterations feed the flow back through the correlation lookup, so round-off is amplified
    iteration by iteration; the growth is measured (1 / 5 / 10 / 20 iterations, printed) and bounded relative to the flow range."""
    from fgt_amd import ops
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", prec)
    sd = _sd("raft_state_keys.json")
    m = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    g = torch.Generator().manual_seed(18)
    base = F.interpolate(torch.rand(1, 3, 61, 109, generator=g), size=(488, 872), mode="bilinear", align_corners=False) * 255
    i1, i2 = base[:, :, 4:484, 4:868].contiguous(), base[:, :, 2:482, 7:871].contiguous()
    errs = {}
    for it in (1, 5, 10, 20):
        ref = RO.raft_forward(sd, i1, i2, iters=it)
        lo, up = m(i1.to(dev), i2.to(dev), iters=it, test_mode=True)
        errs[it] = (rel_err(lo, ref[0]), rel_err(up, ref[1]), ref[1].abs().max().item())
    print(f"[parity] RAFT 864x480 {prec}: rel. error of (flow_low, flow_up) and max |flow_up| by iteration count: "
          + ", ".join(f"{it}: ({a:.2e}, {b:.2e}, {mx:.1f} px)" for it, (a, b, mx) in errs.items()))
    assert errs[1][1] < (5e-5 if prec == "fp32" else 2e-4)
    tol20 = 3e-3 if prec == "fp32" else 1.2e-2      # measured (profiles/r02_run1_pytest_gpu.log): 7.4e-4 / 3.0e-3 on flows of 610 px
    assert errs[20][0] < tol20 and errs[20][1] < tol20


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_raft_864x480_twenty_iterations_contractive_regime_abs_1e3(prec, dev, monkeypatch):
    """VERDICT r2 #4(ii).  With N(0, sigma) weights the 20-iteration GRU loop above is EXPANSIVE (flows of 600 px, a 1e-6 perturbation of the
    weights moves the result by 0.7 px in the oracle itself), so that test measures chaos, not arithmetic.  Here the flow head's last conv
    is scaled by 0.02 — the update per iteration is a fraction of a pixel, flows reach ~10 px after 20 iterations, and the oracle's own
    sensitivity to a 1e-6 relative weight perturbation stays at 8e-5 px (measured with oracle/raft_oracle.py: growth 5x over the loop
    instead of 4000x): the regime of a trained RAFT.  There the north star's 1e-3 ABSOLUTE bar must hold for the exact-fp32 kernels and
    for bf16x3 (16 significant bits per operand) alike."""
    from fgt_amd import ops
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", prec)
    sd = _sd("raft_state_keys.json")
    for k in ("update_block.flow_head.conv2.weight", "update_block.flow_head.conv2.bias"):
        sd[k] = sd[k] * 0.02
    m = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    g = torch.Generator().manual_seed(18)
    base = F.interpolate(torch.rand(1, 3, 61, 109, generator=g), size=(488, 872), mode="bilinear", align_corners=False) * 255
    i1, i2 = base[:, :, 4:484, 4:868].contiguous(), base[:, :, 2:482, 7:871].contiguous()
    ref = RO.raft_forward(sd, i1, i2, iters=20)
    lo, up = m(i1.to(dev), i2.to(dev), iters=20, test_mode=True)
    e_lo, e_up = max_err(lo, ref[0]), max_err(up, ref[1])
    print(f"[parity] RAFT 864x480 / 20 it, contractive regime, {prec}: max abs error flow_low {e_lo:.2e} px, flow_up {e_up:.2e} px "
          f"(flows up to {ref[1].abs().max().item():.1f} px; bar 1e-3 px)")
    assert ref[1].abs().max().item() > 2.0            # the loop does move the flow: not a trivially small signal
    assert e_lo < 1e-3 and e_up < 1e-3


# ------------------------------------------------------------------------------------------------ the bench's batch sizes
@pytest.mark.parametrize("prec", ["bf16x3"])
def test_flow_stages_at_bench_batch_sizes_equal_small_batches(prec, dev, monkeypatch):
    """bench_stages runs RAFT at 64 pairs per refinement batch and LAFC at 16 pivots per call (flow_pipeline.raft_pair_batch /
    lafc_pivot_batch at 432x240); the other flow tests run batches of 3-4.  Per-row results of every kernel are independent of how many
    rows a launch carries, so the fields must be IDENTICAL: asserted here on a 432x240 clip of 34 frames (66 pairs: one full batch of 64 +
    a ragged one) instead of claimed from A/B runs."""
    from fgt_amd import flow_pipeline, ops
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", prec)
    monkeypatch.setattr(ops, "DEFAULT_ATTN_PRECISION", prec)
    H, W, N = 240, 432, 34
    assert flow_pipeline.raft_pair_batch(H, W) == 64 and flow_pipeline.lafc_pivot_batch(H, W) == 16
    raft = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
    raft.load_state_dict(_sd("raft_state_keys.json"), strict=True)
    raft = raft.to(dev)
    lafc = lafc_model.Model(dict(lafc_model.DEFAULT_CONFIG)).eval()
    lafc.load_state_dict(_sd("lafc_vanilla_state_keys.json"), strict=True)
    lafc = lafc.to(dev)
    g = torch.Generator().manual_seed(21)
    base = F.avg_pool2d(torch.rand(1, 3, H + 16, W + 2 * N + 16, generator=g), 7, 1, 3)
    frames = torch.cat([base[:, :, 8:8 + H, 8 + 2 * i:8 + 2 * i + W] for i in range(N)], 0).contiguous().to(dev) * 255.0
    fw64, bw64 = flow_pipeline.compute_flows(raft, frames, iters=4)                      # the bench's batch (64) and encoder batch (16)
    fw4, bw4 = flow_pipeline.compute_flows(raft, frames, iters=4, batch=4, enc_batch=3)
    assert torch.isfinite(fw64).all() and fw64.abs().max() > 0
    assert torch.equal(fw64, fw4) and torch.equal(bw64, bw4)
    flows = fw64.permute(1, 0, 2, 3)[None].contiguous()
    masks = torch.zeros(1, 1, N - 1, H, W, device=dev)
    for i in range(N - 1):
        masks[0, 0, i, 60 + i:150 + i, 100 + 2 * i:260 + 2 * i] = 1
    dif = flow_pipeline.diffusion(flows, masks)
    c16 = flow_pipeline.complete_flows(lafc, flows, masks, dif)                          # 16 pivots per call
    c4 = flow_pipeline.complete_flows(lafc, flows, masks, dif, batch=4)
    assert torch.isfinite(c16).all()
    assert torch.equal(c16, c4)
    # the precomputed-bounds path of the fill (no host read-back in the call) == the default path
    b = ops.hole_bounds(masks[0, 0])
    assert b is not None
    assert torch.equal(flow_pipeline.diffusion(flows, masks, bounds=b), dif)
    rep = ops.solver_report("laplace_fill")
    print(f"[parity] RAFT batch 64 == batch 4 and LAFC 16 == 4 pivots at {W}x{H} ({prec}): identical; fill {rep}")
    assert rep["solver"] == "onchip" and rep["nan_filled"] == 0


# ---- round 6 (ABI 8): the pair batch's correlation volumes as one batched GEMM; z || r of a GRU half as one two-headed conv
@pytest.mark.parametrize("B,n", [(3, 6480), (2, 320), (5, 1624)])
def test_batched_correlation_gemm_equals_per_pair_gemm(B, n, dev, monkeypatch):
    """RAFT/corr.py:52-60: corr[b] = fmap1[b] . fmap2[b]^T / 16.  ops.batched_gemm_nt (fgt_conv_desc.gb_*: group b multiplies its own rows with the
    other frame's feature map AS IT LIES — an interleaved split tensor is a valid weight image) against the per-pair fgt_conv2d GEMM with packed
    weights: the same bf16x3 products in the same order — bit for bit; and against fp64 on the fp32 inputs within the bf16x3 bound."""
    from fgt_amd import ops
    from fgt_amd.ops import PackedConv
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", "bf16x3")
    g = torch.Generator().manual_seed(B * 1000 + n)
    f1, f2 = torch.randn(B, n, 256, generator=g).to(dev), torch.randn(B, n, 256, generator=g).to(dev)
    vol = torch.full((B, n, n), float("nan"), device=dev)
    ops.batched_gemm_nt(ops.split(f1.view(B * n, 256), interleave=True, h=False).view(B, n, 256),
                        ops.split(f2.view(B * n, 256), interleave=True, h=False).view(B, n, 256), vol, scale=1.0 / 16.0)
    ref = torch.empty_like(vol)
    for b in range(B):
        ops.linear(f1[b], PackedConv(f2[b], None), out=ref[b], out_scale=1.0 / 16.0)
    assert torch.equal(vol, ref)
    r64 = torch.matmul(f1[0].double(), f2[0].double().t()) / 16.0
    assert (vol[0].double() - r64).abs().max().item() < 2e-5 * r64.abs().max().item()
    # every wide tile: same values
    for t in ops.BGEMM_CANDIDATES:
        monkeypatch.setattr(ops, "_bgemm_tiles", {(B, n, n, 256): ops.TILE[t]})
        v2 = torch.full((B, n, n), float("nan"), device=dev)
        ops.batched_gemm_nt(ops.split(f1.view(B * n, 256), interleave=True, h=False).view(B, n, 256),
                            ops.split(f2.view(B * n, 256), interleave=True, h=False).view(B, n, 256), v2, scale=1.0 / 16.0)
        assert torch.equal(v2, vol), t


@pytest.mark.parametrize("k,pad", [((1, 5), (0, 2)), ((5, 1), (2, 0))])
def test_two_headed_conv_equals_two_convs(k, pad, dev, monkeypatch):
    """fgt_conv_desc.dual_n0: [z | r] output channels in one launch, head 0 -> sigmoid as fp32, head 1 -> sigmoid * h as a split tensor, with a
    bias MAP and two sources (RAFT's GRU: update.py:46-49, 53-56).  Against the two separate launches: torch.equal (the tap family's tiles are
    bit-identical to each other; columns are independent)."""
    from fgt_amd import ops
    from fgt_amd.ops import PackedConv
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", "bf16x3")
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 30, 54
    rows = B * H * W
    h = torch.randn(rows, 128, generator=g).to(dev)
    m = torch.randn(rows, 128, generator=g).to(dev)
    wz, wr = (torch.randn(128, 256, *k, generator=g) * 0.03).to(dev), (torch.randn(128, 256, *k, generator=g) * 0.03).to(dev)
    bm = torch.randn(rows, 256, generator=g).to(dev)
    hs, ms = ops.split(h, h=False), ops.split(m, h=False)
    v4 = lambda s: s.view(B, H, W, 128)
    z_ref = ops.conv2d(v4(hs), PackedConv(wz, None), x1=v4(ms), bias_map=bm[:, :128].contiguous(), pad=pad, act="sigmoid")
    rh_ref = ops.conv2d(v4(hs), PackedConv(wr, None), x1=v4(ms), bias_map=bm[:, 128:].contiguous(), pad=pad, act="sigmoid", epi="mul", aux1=h, out_split="only")
    z, rh = ops.conv2d(v4(hs), PackedConv(torch.cat([wz, wr], 0), None), x1=v4(ms), bias_map=bm, pad=pad, act="sigmoid", epi="mul", aux1=h,
                       out_split="both", dual=True)
    # (the tap kernels build the two-headed / bias-map epilogues for their 5-tap instances only — RAFT's GRU convs; a 3 x 3 layer with them is declined)
    with pytest.raises(RuntimeError, match="5 reused taps"):
        w3 = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev)
        ops.conv2d(v4(hs), PackedConv(w3, None), x1=v4(ms), bias_map=bm, pad=1, act="sigmoid", epi="mul", aux1=h, out_split="both", dual=True, tile="128x128t")
    assert z.shape == (B, H, W, 128) and rh.shape[-1] == 128
    assert torch.equal(z, z_ref)
    assert torch.equal(rh.data, rh_ref.data)
    zt = torch.sigmoid(F.conv2d(torch.cat([h, m], 1).view(B, H, W, 256).permute(0, 3, 1, 2).double(), wz.double(), None, 1, pad).permute(0, 2, 3, 1) + bm[:, :128].view(B, H, W, 128).double())
    assert (z.double() - zt).abs().max().item() < 1e-5


@pytest.mark.parametrize("prec", ["bf16x3"])
def test_raft_fused_zr_and_batched_correlation_equal_the_separate_launches(prec, dev, monkeypatch):
    from fgt_amd import ops
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", prec)
    g = load_golden("raft_128x160_it6.npz")
    outs = {}
    for new in (True, False):
        m = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
        m.load_state_dict(_sd("raft_state_keys.json"), strict=True)
        m = m.to(dev)
        m.fuse_zr = m.batched_corr = new
        outs[new] = m(g["image1"].to(dev), g["image2"].to(dev), iters=6, test_mode=True)
        assert report(f"raft flow_up zr/bcorr={new} {prec}", outs[new][1], g["flow_up"])[1] < 1e-3
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])


def test_lookup_coordinate_table_and_deep_staging_are_bit_identical_to_the_per_tap_loop(dev):
    """Round 6: fgt_corr_lookup evaluates the (separable) sample coordinates once per (pixel, level, axis, tap index) and requests a thread's whole share of
    the windows before using it.  Same expressions, same order: the taps must equal the per-tap loop's bit for bit (child processes: the switches are
    read once per process), incl. coordinates far outside the map and a ragged last block."""
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, ".")
from fgt_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
B, H1, W1 = 3, 30, 54
rows = B * H1 * W1 - 5
vol = torch.randn(B * H1 * W1, H1, W1, generator=g).to(dev)
pyr, hh, ww = [vol], H1, W1
for _ in range(3):
    pyr.append(ops.avgpool2(pyr[-1], B * H1 * W1, hh, ww)); hh, ww = hh // 2, ww // 2
ys, xs = torch.meshgrid(torch.arange(H1), torch.arange(W1), indexing="ij")
coords = (torch.stack([xs, ys], -1).float().unsqueeze(0).repeat(B, 1, 1, 1) + torch.randn(B, H1, W1, 2, generator=g) * 20.0).contiguous().to(dev)
out = torch.empty(B, H1, W1, 324, device=dev)
ops.corr_lookup(pyr, B, H1, W1, 4, coords, out)
torch.save(out.cpu(), sys.argv[1])
'''
    import os
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({"FGT_LOOKUP_TABLE": "0", "FGT_LOOKUP_DEEP": "0"}, {"FGT_LOOKUP_TABLE": "1", "FGT_LOOKUP_DEEP": "1"}):
        f = tempfile.NamedTemporaryFile(suffix=".pt", delete=False).name
        r = subprocess.run([sys.executable, "-c", code, f], cwd=root, env=dict(os.environ, **env), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(f))
        os.remove(f)
    assert torch.equal(outs[0], outs[1])
