"""End-to-end parity of the MI355X FGT forward against (a) golden vectors produced by the reference itself and
(b) the CPU oracle on the same seeded inputs.  Bar (BASELINE.json north_star): max |diff| <= 1e-3 in fp32; the
fp32-MFMA path is expected to land around 1e-6, so the tests also bound the error relative to the output scale."""
import json
import os

import pytest
import torch

from fgt_amd.fgt_model import DEFAULT_CONFIG, Model
from fgt_amd.synth import synth_state_dict
from oracle import fgt_oracle as O
from util import GOLDEN, fgt_inputs, load_golden, report

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

ABS_TOL = 1e-3   # north-star bar
REL_TOL = 2e-4   # tighter: relative to max |reference output| (random-init outputs are only ~0.09)


def _model(dev, conv_type="vanilla", seed=0, mode="normal"):
    m = Model(dict(DEFAULT_CONFIG, conv_type=conv_type)).eval()
    sd = synth_state_dict(m.state_dict(), seed=seed, mode=mode)
    m.load_state_dict(sd, strict=True)
    return m.to(dev), sd


@pytest.mark.parametrize("name", ["fgt_vanilla_64x96x3.npz", "fgt_vanilla_48x80x3.npz"])
def test_fgt_forward_matches_reference_golden(name, dev):
    g = load_golden(name)
    m, _ = _model(dev)
    out = m(g["masked_frames"].to(dev), g["flows"].to(dev), g["masks"].to(dev))
    e, r = report(name, out, g["out"])
    assert e < ABS_TOL and r < REL_TOL


def test_fgt_gated_matches_reference_golden(dev):
    g = load_golden("fgt_gated_48x64x2.npz")
    m, _ = _model(dev, "gated")
    out = m(g["masked_frames"].to(dev), g["flows"].to(dev), g["masks"].to(dev))
    e, r = report("gated", out, g["out"])
    assert e < ABS_TOL and r < REL_TOL


def test_fgt_trained_grid_240x432_matches_reference_golden(dev):
    g = load_golden("fgt_vanilla_240x432x2.npz")
    mf, fl, ms = fgt_inputs(240, 432, 2, 14)
    m, _ = _model(dev)
    out = m(mf.to(dev), fl.to(dev), ms.to(dev))
    e, r = report("240x432x2", out, g["out"])
    assert e < ABS_TOL and r < REL_TOL


def test_fgt_blocks_match_oracle_with_order_one_activations(dev):
    """Single temporal / spatial blocks on N(0,1) tokens (BASELINE config #2 shape, fp32): block output is
    residual dominated (max ~5), so this checks absolute accuracy at O(1) scale."""
    m, sd = _model(dev)
    net = m.net
    P = net.packed()
    t, th, tw = 3, 20, 36
    g = torch.Generator().manual_seed(5)
    x = torch.randn(t * th * tw, 512, generator=g)
    f = torch.randn(t * th * tw, 256, generator=g)
    ref_s = O.spatial_block(x.view(t, -1, 512), f.view(t, -1, 256), sd, "net.first_s_transformer.", th, tw, (60, 108))
    out_s = net._spatial(x.to(dev), f.to(dev), P["s0"], t, th, tw, 60, 108)
    e, r = report("spatial block", out_s.view(t, -1, 512), ref_s)
    assert e < 1e-4
    ref_t = O.temporal_block(x.view(t, -1, 512), sd, "net.first_t_transformer.", t, th, tw, (60, 108))
    out_t = net._temporal(x.to(dev), P["t0"], 1, t, th, tw, 60, 108)
    e, r = report("temporal block", out_t.view(t, -1, 512), ref_t)
    assert e < 1e-4


def test_fgt_batch_of_two_clips_matches_oracle(dev):
    m, sd = _model(dev)
    a = fgt_inputs(48, 64, 2, 21)
    b = fgt_inputs(48, 64, 2, 22)
    mf, fl, ms = (torch.cat([u, v], 0) for u, v in zip(a, b))
    out = m(mf.to(dev), fl.to(dev), ms.to(dev))
    ref = O.fgt_forward(sd, DEFAULT_CONFIG, mf, fl, ms)
    e, r = report("batch 2", out, ref)
    assert e < ABS_TOL and r < REL_TOL


def test_outputs_are_fresh_and_inputs_untouched(dev):
    m, _ = _model(dev)
    mf, fl, ms = (x.to(dev) for x in fgt_inputs(48, 64, 2, 23))
    keep = (mf.clone(), fl.clone(), ms.clone())
    o1 = m(mf, fl, ms)
    o2 = m(mf, fl, ms)
    assert torch.equal(o1, o2) and o1.data_ptr() != o2.data_ptr()
    assert all(torch.equal(a, b) for a, b in zip(keep, (mf, fl, ms)))
    assert o1.shape == (2, 3, 48, 64) and o1.device.type == "cuda"


def test_fgt_bf16x3_conv_precision_within_fp32_bar(dev, monkeypatch):
    """Same forward with every conv/GEMM product issued as 3 bf16 MFMAs on hi/lo splits: must still meet the 1e-3 bar
    (and stay ~1e-5 relative): this is the fast path bench.py can select with --precision bf16x3."""
    from fgt_amd import ops
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", "bf16x3")
    monkeypatch.setattr(ops, "DEFAULT_ATTN_PRECISION", "bf16x3")
    g = load_golden("fgt_vanilla_240x432x2.npz")
    mf, fl, ms = fgt_inputs(240, 432, 2, 14)
    m, _ = _model(dev)
    out = m(mf.to(dev), fl.to(dev), ms.to(dev))
    e, r = report("240x432x2 bf16x3", out, g["out"])
    assert e < ABS_TOL and r < 2e-3
    g2 = load_golden("fgt_vanilla_48x80x3.npz")
    out = m(g2["masked_frames"].to(dev), g2["flows"].to(dev), g2["masks"].to(dev))
    e, r = report("48x80x3 bf16x3", out, g2["out"])
    assert e < ABS_TOL and r < 2e-3


@pytest.mark.parametrize("H,W,t", [(256, 432, 2), (480, 864, 2)])
def test_fgt_inference_grids_match_oracle(H, W, t, dev):
    """Token grids other than the trained 20x36: the tool's default 256x432 (22x36 tokens, spatial padding to 24x40) and
    BASELINE config #5's 864x480 (40x72 tokens, 45 windows, 180 global tokens) — the reference's `inference` branches."""
    m, sd = _model(dev)
    mf, fl, ms = fgt_inputs(H, W, t, 31)
    ref = O.fgt_forward(sd, DEFAULT_CONFIG, mf, fl, ms)
    out = m(mf.to(dev), fl.to(dev), ms.to(dev))
    e, r = report(f"fgt {W}x{H}x{t} fp32", out, ref)
    assert e < ABS_TOL and r < REL_TOL


@pytest.mark.parametrize("prec,blk_tol,att_tol", [("fp32", 2e-5, 2e-6), ("bf16x3", 1e-3, 1e-4)])
def test_config2_spatial_block_t10_order_one_tokens(prec, blk_tol, att_tol, dev, monkeypatch):
    """BASELINE config #2 as specified: ONE SpatialTransformer (token grid 20x36, 512 / 256 channels, 4 heads,
    window 8, global down-size 4, mlp 40), t = 10, x, f ~ N(0,1), reference-style N(0, 0.02) weights.
    SURVEY §7 measured what plain bf16 OPERANDS (fp32 accumulate) do in this regime: 2.3e-3 on the block output (the FFN GEMMs at
    O(1) activations dominate), 4.5e-4 on the attention branch (max 0.085) — i.e. they miss the 1e-3 bar.  The bf16x3 mode
    (hi/lo split, 3 MFMAs per product) must hold it with margin: block (max ~5, residual dominated) < 1e-3 absolute, attention
    branch (pre-residual) < 1e-4."""
    from fgt_amd import ops
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", prec)
    monkeypatch.setattr(ops, "DEFAULT_ATTN_PRECISION", prec)
    m, sd = _model(dev)
    net = m.net
    P = net.packed()
    t, th, tw = 10, 20, 36
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(t * th * tw, 512, generator=g)
    f = torch.randn(t * th * tw, 256, generator=g)
    p = "net.first_s_transformer."
    ref_att = O.swmhsa(x.view(t, -1, 512), f.view(t, -1, 256), sd, p + "attention.", th, tw)
    ref_blk = O.spatial_block(x.view(t, -1, 512), f.view(t, -1, 256), sd, p, th, tw, (60, 108))
    xd, fd = x.to(dev), f.to(dev)
    att = (net._spatial_attention(xd, fd, P["s0"], t, th, tw) - xd).cpu().view(t, -1, 512)
    blk = net._spatial(xd, fd, P["s0"], t, th, tw, 60, 108).cpu().view(t, -1, 512)
    ea, _ = report(f"C2 attention branch {prec} (ref max {ref_att.abs().max().item():.3f})", att, ref_att)
    eb, _ = report(f"C2 spatial block {prec} (ref max {ref_blk.abs().max().item():.2f})", blk, ref_blk)
    # the attention branch is recovered as (x + a) - x in fp32: that subtraction alone costs ~ulp(5) = 5e-7
    assert ea < att_tol and eb < blk_tol


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_config2_spatial_mhsa_three_streams_equal_one_stream_and_graph_replay(prec, dev, monkeypatch):
    """The C2-sized SWMHSA call runs its value / key / query paths on three HIP streams (fgt_model.SPATIAL_STREAM_ROWS): the same kernels on
    the same data — bit-identical to the single-stream order, repeatably (20 calls), and as a captured hipGraph (fork / join inside the capture)."""
    from fgt_amd import fgt_model, ops
    from fgt_amd.graph import GraphedCall
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", prec)
    monkeypatch.setattr(ops, "DEFAULT_ATTN_PRECISION", prec)
    m, sd = _model(dev)
    net = m.net
    P = net.packed()
    t, th, tw = 10, 20, 36
    g = torch.Generator().manual_seed(99)
    x, f = torch.randn(t * th * tw, 512, generator=g).to(dev), torch.randn(t * th * tw, 256, generator=g).to(dev)
    monkeypatch.setattr(fgt_model, "SPATIAL_STREAM_ROWS", 0)
    one = net._spatial_attention(x, f, P["s0"], t, th, tw).clone()
    monkeypatch.setattr(fgt_model, "SPATIAL_STREAM_ROWS", 32768)
    monkeypatch.setattr(fgt_model, "SPATIAL_STREAMS_EAGER", True)
    for _ in range(20):
        assert torch.equal(net._spatial_attention(x, f, P["s0"], t, th, tw), one)
    gc = GraphedCall(lambda a, b: net._spatial_attention(a, b, P["s0"], t, th, tw), [x, f])
    for _ in range(5):
        assert torch.equal(gc(x, f), one)


def test_config2_temporal_block_t10_bf16x3(dev, monkeypatch):
    """The temporal counterpart at the same shape (TMHSA t = 10: zone length 1800, below the 8-wavefront switch)."""
    from fgt_amd import ops
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", "bf16x3")
    monkeypatch.setattr(ops, "DEFAULT_ATTN_PRECISION", "bf16x3")
    m, sd = _model(dev)
    net = m.net
    t, th, tw = 10, 20, 36
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(t * th * tw, 512, generator=g)
    ref = O.temporal_block(x.view(t, -1, 512), sd, "net.first_t_transformer.", t, th, tw, (60, 108))
    out = net._temporal(x.to(dev), net.packed()["t0"], 1, t, th, tw, 60, 108)
    assert report("C2-shaped temporal block bf16x3", out.view(t, -1, 512), ref)[0] < 1e-3
