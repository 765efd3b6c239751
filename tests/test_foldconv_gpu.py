"""nn.Fold behind a per-token Linear as ONE token-grid convolution (fgt_conv_desc.ps_r, ABI 7) on a real MI355X.

FusionFeedForward's first half (ffn_base.py:53-66: conv1 -> fold / fold(ones) -> ReLU) and Vec2Patch + encoder residual
(model.py:102-110, 280) run, in the bf16x3 mode, as a 3x3 stride-1 convolution over the token grid whose epilogue scatters the
(ry, rx, c) columns of a token cell to the pixels (3I + ry, 3J + rx) of the folded map.  Gates: (i) distance to an fp64 evaluation of the
REFERENCE formulation (Linear, F.fold, division by F.fold(ones)) on the same split operands; (ii) bit equality among the tap tiles and
between ky-skipping on / off (the skipped products are exact zeros); (iii) the model with the fold convolution on and off against the
reference golden output."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

TAPS = ["128x128x8t", "128x128t", "128x64t", "128x64x8t", "64x64t", "128x128it", "256x128it", "256x256it"]


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _hi_lo(t):
    """The value a bf16x3 kernel sees for an fp32 operand: bf16_rne(x) + bf16_rne(x - hi), as fp64."""
    hi = t.to(torch.bfloat16)
    return (hi.float() + (t - hi.float()).to(torch.bfloat16).float()).double()


def _reference(x, w, b, N, th, tw, cc, k, s, p, Hf, Wf, normalize, res=None, relu=True):
    """fp64: Linear -> nn.Fold (-> / fold(ones)) (+ residual) (-> ReLU), channels-last result [N, Hf, Wf, cc]."""
    y = _hi_lo(x).view(N, th * tw, -1) @ _hi_lo(w).t() + b.double()                    # [N, n, cc*k*k], column c*k*k + ky*k + kx
    f = F.fold(y.permute(0, 2, 1), (Hf, Wf), k, stride=s, padding=p)
    if normalize:
        f = f / F.fold(torch.ones(N, k * k, th * tw, dtype=torch.float64, device=x.device), (Hf, Wf), k, stride=s, padding=p)
    f = f.permute(0, 2, 3, 1)
    if res is not None:
        f = f + res.double()
    return F.relu(f) if relu else f


GEOMS = [
    # name, N, th, tw, Hf, Wf, cin, cc
    ("ffn_6x8_cut_map", 3, 6, 8, 16, 24, 64, 40),          # 64x96 input: 3*6 = 18 > 16 rows — the last token row's sub-pixels 1, 2 fall off the map
    ("ffn_4x7_cut_cols", 2, 4, 7, 12, 20, 96, 40),          # 48x80 input: 3*7 = 21 > 20 columns
    ("ffn_20x36_bench_grid", 2, 20, 36, 60, 108, 512, 40),  # the benchmark's token grid and channel counts
    ("ffn_22x36_tool_default_height", 1, 22, 36, 64, 108, 128, 40),   # 256 x 432 input (the tool's default): 22 x 3 = 66 > 64 feature rows
    ("ffn_1x1_grid", 5, 1, 1, 3, 3, 32, 8),
    ("v2p_5x9_c128", 2, 5, 9, 15, 26, 64, 128),             # Vec2Patch: 128 channels per pixel (g0 = 384: no padding columns)
]


@pytest.mark.parametrize("geom", GEOMS, ids=[g[0] for g in GEOMS])
@pytest.mark.parametrize("il", [False, True], ids=["planes", "interleaved"])
def test_fold_conv_vs_reference_formulation(geom, il, dev):
    from fgt_amd import fgt_model as M, ops
    name, N, th, tw, Hf, Wf, cin, cc = geom
    k, s, p = 7, 3, 3
    x = _rand(N * th * tw, cin, seed=1).to(dev)
    w = _rand(cc * k * k, cin, seed=2, scale=1.0 / math.sqrt(cin)).to(dev)
    b = _rand(cc * k * k, seed=3, scale=0.3).to(dev)
    g0, cout, _ = M.fold_conv_layout(cc, s)
    pc = ops.PackedConv(M.fold_conv_weight(w, cc, k, s), None)
    xs = ops.split(x, interleave=il).view(N, th, tw, cin)
    for normalize in (True, False):
        res = None if normalize else _rand(N, Hf, Wf, cc, seed=4).to(dev)
        ref = _reference(x, w, b, N, th, tw, cc, k, s, p, Hf, Wf, normalize, res, relu=normalize)
        scale = max(ref.abs().max().item(), 1.0)
        off, sc = M.fold_conv_tables(b, cc, k, s, th, tw, normalize)
        off, sc = off.to(dev), (None if sc is None else sc.to(dev))
        kw = dict(stride=1, pad=1, aux_per_image=True, ps=(s, cc, g0, Hf, Wf), precision="bf16x3")
        kw.update(dict(act="relu", epi="affine", aux1=off, aux2=sc) if normalize else dict(epi="ps_add2", aux1=off, aux2=res))
        first = None
        for t in TAPS + ["128x128", "128x128x8ea"]:
            outs = {}
            for skip in (0, g0):
                o32, osp = ops.conv2d(xs, pc, tile=t, ky_skip_n0=skip, out_split="both", **kw)
                torch.cuda.synchronize()
                assert tuple(o32.shape) == (N, Hf, Wf, cc)
                assert torch.equal(osp.data, ops.split(o32).data), f"{name} {t}: split output != split(fp32 output)"
                outs[skip] = o32
            assert torch.equal(outs[0], outs[g0]), f"{name} {t}: skipping the all-zero ky = 0 taps changed the result"
            e = (outs[g0].double() - ref).abs().max().item()
            assert e <= 2e-5 * scale, f"{name} {t} normalize={normalize}: {e:.3e} (scale {scale:.2e})"
            if t in TAPS:
                first = outs[g0] if first is None else first
                assert torch.equal(outs[g0], first), f"{name}: tap tile {t} differs from {TAPS[0]}"
        auto = ops.conv2d(xs, pc, ky_skip_n0=g0, **kw)
        assert torch.equal(auto, first), "tile = auto: the tap-reusing kernel"
        # N-major tile walk inside an XCD (fgt_conv_desc.tile_order = 1: what the model passes for these layers): the order of tiles, not of any sum
        for t in ("128x128it", "128x64t", "128x128"):
            assert torch.equal(ops.conv2d(xs, pc, tile=t, ky_skip_n0=g0, tile_order=1, **kw), ops.conv2d(xs, pc, tile=t, ky_skip_n0=g0, tile_order=0, **kw)), t
        # the formulation it replaces: Linear (tap-major columns) + fgt_fold on the same operands
        w1p = w.view(cc, k * k, cin).permute(1, 0, 2).reshape(cc * k * k, cin)
        b1p = b.view(cc, k * k).permute(1, 0).reshape(-1)
        Y = ops.linear(ops.split(x, interleave=il), ops.PackedConv(w1p, b1p), precision="bf16x3")
        old = ops.fold(Y, N, th, tw, cc, k, s, p, Hf, Wf, normalize=normalize, res=res, relu=normalize)
        e_old = (old.double() - ref).abs().max().item()
        e_new = (first.double() - ref).abs().max().item()
        assert e_new <= max(3.0 * e_old, 2e-6 * scale)
        print(f"[parity] fold conv {name} ({'interleaved' if il else 'planes'}, normalize={normalize}): max |conv - fp64 reference formulation| {e_new:.2e}, "
              f"|Linear + fgt_fold - fp64| {e_old:.2e} (values up to {scale:.2f})")


def test_fold_conv_interleaved_split_output(dev):
    """Vec2Patch hands the decoder an INTERLEAVED split map (128 channels per pixel): the sub-pixel epilogue writes channel c of a pixel at
    (c / 32) * 64 + c % 32."""
    from fgt_amd import fgt_model as M, ops
    N, th, tw, Hf, Wf, cin, cc, k, s = 2, 5, 9, 15, 26, 64, 128, 7, 3
    x = _rand(N * th * tw, cin, seed=1).to(dev)
    w = _rand(cc * k * k, cin, seed=2, scale=0.1).to(dev)
    b = _rand(cc * k * k, seed=3, scale=0.3).to(dev)
    res = _rand(N, Hf, Wf, cc, seed=4).to(dev)
    g0 = M.fold_conv_layout(cc, s)[0]
    pc = ops.PackedConv(M.fold_conv_weight(w, cc, k, s), None)
    off = M.fold_conv_tables(b, cc, k, s, th, tw, False)[0].to(dev)
    xs = ops.split(x, interleave=True).view(N, th, tw, cin)
    kw = dict(stride=1, pad=1, epi="ps_add2", aux1=off, aux2=res, aux_per_image=True, ps=(s, cc, g0, Hf, Wf), ky_skip_n0=g0, precision="bf16x3")
    o32 = ops.conv2d(xs, pc, **kw)
    osp = ops.conv2d(xs, pc, out_split="only", out_il=True, **kw)
    assert osp.il and torch.equal(osp.data, ops.split(o32, interleave=True).data)


@pytest.mark.parametrize("name,conv_type", [("fgt_vanilla_64x96x3.npz", "vanilla"), ("fgt_vanilla_48x80x3.npz", "vanilla"), ("fgt_gated_48x64x2.npz", "gated")])
def test_model_with_and_without_fold_conv_vs_reference_golden(name, conv_type, dev, monkeypatch):
    """The bf16x3 model with the fold convolution (default) and with Linear + fgt_fold (FGT_FOLD_CONV=0) against the REFERENCE's output on
    inputs whose token grid does not tile the feature map exactly (64x96: 6x8 tokens over 16x24; 48x80: 4x7 over 12x20)."""
    from fgt_amd import fgt_model as M, ops
    from fgt_amd.synth import synth_state_dict
    from util import load_golden, max_err
    g = load_golden(name)
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", "bf16x3")
    monkeypatch.setattr(ops, "DEFAULT_ATTN_PRECISION", "bf16x3")
    outs = {}
    for fc in (True, False):
        monkeypatch.setattr(M, "FOLD_CONV", fc)
        m = M.Model(dict(M.DEFAULT_CONFIG, conv_type=conv_type)).eval()
        m.load_state_dict(synth_state_dict(m.state_dict(), seed=0), strict=True)
        m = m.to(dev)
        assert ("fc" in m.net.packed()["t0"]["ffn"]) == fc
        outs[fc] = m(g["masked_frames"].to(dev), g["flows"].to(dev), g["masks"].to(dev)).cpu()
    e_new, e_old = max_err(outs[True], g["out"]), max_err(outs[False], g["out"])
    print(f"[parity] FGT bf16x3 {name}: fold conv max_abs={e_new:.3e}, Linear + fold max_abs={e_old:.3e}, between them {max_err(outs[True], outs[False]):.3e} "
          f"(ref max {g['out'].abs().max().item():.3e})")
    assert e_new < 5e-5 and e_old < 5e-5
