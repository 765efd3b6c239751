"""Pre-split operand path (fgt_conv_desc.in_split / out_split, fgt_split, LDS-DMA conv kernel) on a real MI355X.

The split path is the bf16x3 arithmetic with different data movement, so its gate is BIT equality with the register-staged
bf16x3 kernel on the same inputs (same hi/lo values, same MFMA order), which is itself held to the fp32 reference in
test_ops_gpu.py::test_conv2d_bf16x3_split_precision."""
import math

import pytest
import torch
import torch.nn.functional as F

from util import report

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

TILES = ["128x128", "128x64", "64x64", "128x32", "256x128", "128x128x8", "256x128x16", "256x64x8", "128x128ea", "128x64ea", "64x64ea", "128x128x8ea", "256x128x16ea", "256x64x8ea"]
# (the ring / ping-pong / interleaved-schedule / 8-phase / loader-wavefront variants of round 1-2 measured within +-5 % or slower and now exist in
#  diagnostic builds only: csrc/diag/conv_split_variants.hip, `fgt_amd.build.build(variant="diag")`)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def test_split_format(dev):
    """fgt_split: hi = bf16_rne(x), lo = bf16_rne(x - hi); relu flag; strided source."""
    from fgt_amd import ops
    x = _rand(333, 72, seed=1, scale=3.0)
    x[0, :4] = torch.tensor([0.0, -0.0, 1e-30, 65504.0])
    s = ops.split(x.to(dev))
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    assert torch.equal(s.data[0].cpu(), hi) and torch.equal(s.data[1].cpu(), lo)
    assert (s.float().cpu() - x).abs().max() <= x.abs().max() * 2.0 ** -16
    xr = x.clamp_min(0)
    sr = ops.split(x.to(dev), relu=True)
    assert torch.equal(sr.data[0].cpu(), xr.to(torch.bfloat16))
    big = _rand(70000, 256, seed=3).to(dev)            # > 2^22 float4s: more work items than one grid pass
    sb = ops.split(big)
    assert torch.equal(sb.data[0], big.to(torch.bfloat16)) and torch.equal(sb.data[1], (big - sb.data[0].float()).to(torch.bfloat16))
    wide = _rand(50, 96, seed=2).to(dev)
    sv = ops.split(wide[:, 16:48])                     # channel slice of a wider buffer
    assert torch.equal(sv.data[0].cpu(), wide[:, 16:48].cpu().to(torch.bfloat16))


SPLIT_CASES = [
    # name, N, H, W, Cin, Cout, k, stride, pad, dil, groups
    ("3x3_s1", 2, 20, 28, 64, 128, 3, 1, 1, 1, 1),
    ("3x3_s2", 2, 24, 40, 64, 64, 3, 2, 1, 1, 1),
    ("3x3_cout_odd", 1, 17, 23, 32, 126, 3, 1, 1, 1, 1),
    ("7x7_s3_p3_cin40", 2, 24, 36, 40, 96, 7, 3, 3, 1, 1),      # K = 1960: K tail inside the last step
    ("3x3_dil8_cin48", 1, 30, 27, 48, 48, 3, 1, 8, 8, 1),
    ("1x1_linear", 1, 1, 700, 512, 1960, 1, 1, 0, 1, 1),
    ("1x1_k8", 1, 1, 130, 8, 40, 1, 1, 0, 1, 1),                  # a single 8-channel chunk
    ("g4", 1, 15, 27, 64, 96, 3, 1, 1, 1, 4),
    ("1x5", 1, 20, 30, 64, 64, (1, 5), 1, (0, 2), 1, 1),
]


@pytest.mark.parametrize("case", SPLIT_CASES, ids=[c[0] for c in SPLIT_CASES])
@pytest.mark.parametrize("tile", TILES)
def test_conv_split_inputs_bit_equal_to_bf16x3(case, tile, dev):
    from fgt_amd import ops
    name, N, H, W, Cin, Cout, k, s, p, d, g = case
    kh, kw = (k, k) if isinstance(k, int) else k
    x = _rand(N, H, W, Cin, seed=1).to(dev)
    w = _rand(Cout, Cin // g, kh, kw, seed=2, scale=1.0 / math.sqrt(Cin // g * kh * kw))
    b = _rand(Cout, seed=3)
    pc = ops.PackedConv(w.to(dev), b.to(dev), groups=g)
    ref = ops.conv2d(x, pc, stride=s, pad=p, dil=d, act="lrelu", tile="128x128", precision="bf16x3")
    got = ops.conv2d(ops.split(x), pc, stride=s, pad=p, dil=d, act="lrelu", tile=tile, precision="bf16x3")
    torch.cuda.synchronize()
    assert torch.equal(got, ref), f"{name} tile={tile}: max diff {(got - ref).abs().max().item():.3e}"
    if tile == "128x128":      # and the pair is held to fp32 torch
        t = F.leaky_relu(F.conv2d(x.cpu().permute(0, 3, 1, 2), w, b, s, p, d, g), 0.2).permute(0, 2, 3, 1)
        e, r = report(f"split conv {name}", got.cpu(), t)
        assert r < 2e-5


def test_conv_split_two_source_grouped_upsample_replicate(dev):
    """Encoder-style group-interleaved concat of two split sources; nearest-x2 upsample; replicate padding."""
    from fgt_amd import ops
    N, H, W, g = 2, 15, 27, 8
    x0, o = _rand(N, H, W, 256, seed=1).to(dev), _rand(N, H, W, 384, seed=2).to(dev)
    w, b = _rand(256, 640 // g, 3, 3, seed=3, scale=0.05), _rand(256, seed=4)
    pc = ops.PackedConv(w.to(dev), b.to(dev), groups=g)
    ref = ops.conv2d(x0, pc, x1=o, stride=1, pad=1, act="lrelu", precision="bf16x3")
    for tile in ("auto", "64x64", "128x128x8"):
        got = ops.conv2d(ops.split(x0), pc, x1=ops.split(o), stride=1, pad=1, act="lrelu", precision="bf16x3", tile=tile)
        assert torch.equal(got, ref), tile
    x = _rand(1, 12, 20, 32, seed=5).to(dev)
    w2, b2 = _rand(48, 32, 3, 3, seed=6, scale=0.1), _rand(48, seed=7)
    pc2 = ops.PackedConv(w2.to(dev), b2.to(dev))
    for kw in (dict(upsample=True, pad=1), dict(pad=2, pad_mode="replicate", dil=2)):
        ref = ops.conv2d(x, pc2, precision="bf16x3", **kw)
        # (tile = auto would take the 2x2 sub-pixel form of the upsampled layer, another summation: tests/test_up4.py)
        got = ops.conv2d(ops.split(x), pc2, precision="bf16x3", tile="64x64" if kw.get("upsample") else None, **kw)
        assert torch.equal(got, ref), kw


def test_conv_split_channel_slices_and_row_slices(dev):
    """Split sources that are slices of wider / longer buffers (ld > C, leading-dim slices)."""
    from fgt_amd import ops
    rows = 900
    wide = _rand(rows + 40, 96, seed=1).to(dev)
    ws = ops.split(wide)
    w, b = _rand(72, 64, seed=2, scale=0.1), _rand(72, seed=3)
    pc = ops.PackedConv(w.to(dev), b.to(dev))
    ref = ops.linear(wide[8:8 + rows, 16:80], pc, precision="bf16x3")
    sl = ops.Split(ws.data[:, 8:8 + rows, 16:80])
    got = ops.linear(sl, pc, precision="bf16x3")
    assert torch.equal(got, ref)


@pytest.mark.parametrize("mode", ["only", "both"])
def test_conv_out_split_epilogue(mode, dev):
    """out_split: the epilogue's split planes equal fgt_split of the fp32 result; with epilogue combine + activation."""
    from fgt_amd import ops
    x = _rand(2, 18, 26, 64, seed=1).to(dev)
    w, b = _rand(128, 64, 3, 3, seed=2, scale=0.05), _rand(128, seed=3)
    aux = _rand(2, 18, 26, 128, seed=4).to(dev)
    pc = ops.PackedConv(w.to(dev), b.to(dev))
    for xin in (x, ops.split(x)):
        for tile in ("128x128", "128x128x8", "64x64"):
            ref = ops.conv2d(x, pc, pad=1, act="lrelu", epi="add", aux1=aux, precision="bf16x3", tile="128x128")
            r = ops.conv2d(xin, pc, pad=1, act="lrelu", epi="add", aux1=aux, precision="bf16x3", tile=tile, out_split=mode)
            o32, osp = (None, r) if mode == "only" else r
            want = ops.split(ref)
            assert torch.equal(osp.data, want.data)
            if o32 is not None:
                assert torch.equal(o32, ref)
    # a chain conv -> conv entirely in split form equals the fp32-tensor chain
    w2, b2 = _rand(64, 128, 3, 3, seed=5, scale=0.05), _rand(64, seed=6)
    pc2 = ops.PackedConv(w2.to(dev), b2.to(dev))
    a = ops.conv2d(ops.conv2d(x, pc, pad=1, act="lrelu", precision="bf16x3"), pc2, pad=1, stride=2, precision="bf16x3")
    # (explicit tile: a 3x3 / stride-1 layer on split inputs is otherwise routed to the tap-reusing kernel, whose accumulation order differs)
    s1 = ops.conv2d(ops.split(x), pc, pad=1, act="lrelu", precision="bf16x3", out_split="only", tile="128x128")
    bsp = ops.conv2d(s1, pc2, pad=1, stride=2, precision="bf16x3")
    assert torch.equal(a, bsp)


def test_split_rejections(dev):
    from fgt_amd import ops
    x = _rand(1, 8, 8, 36, seed=1).to(dev)            # 36 channels: not a multiple of 8
    pc = ops.PackedConv(_rand(16, 36, 3, 3, seed=2).to(dev), None)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        ops.conv2d(ops.split(x), pc, pad=1, precision="bf16x3", tile="64x64")
    x = _rand(1, 8, 8, 32, seed=1).to(dev)
    pc = ops.PackedConv(_rand(16, 32, 3, 3, seed=2).to(dev), None)
    with pytest.raises(RuntimeError, match="bf16x3"):
        ops.conv2d(ops.split(x), pc, pad=1, precision="fp32")
    with pytest.raises(RuntimeError, match="in_relu"):
        ops.conv2d(ops.split(x), pc, pad=1, precision="bf16x3", in_relu=True, tile="64x64")


# ---- interleaved layout (in_split = 2, w_il): [hi 32 | lo 32] per 32 channels = one 128-byte line per pixel and K-step
IL_CASES = [c for c in SPLIT_CASES if (c[4] // c[10]) % 32 == 0]


def test_split_interleaved_format(dev):
    from fgt_amd import ops
    x = _rand(3, 5, 7, 96, seed=1, scale=2.0).to(dev)
    a, b = ops.split(x), ops.split(x, interleave=True)
    assert b.data.shape == (3, 5, 7, 192) and tuple(b.shape) == (3, 5, 7, 96)
    hi, lo = b.planes()
    assert torch.equal(hi, a.data[0]) and torch.equal(lo, a.data[1])
    assert torch.equal(b.data[..., 0:32], a.data[0][..., 0:32]) and torch.equal(b.data[..., 32:64], a.data[1][..., 0:32])
    assert torch.equal(b.data[..., 64:96], a.data[0][..., 32:64])


@pytest.mark.parametrize("case", IL_CASES, ids=[c[0] for c in IL_CASES])
@pytest.mark.parametrize("tile", ["128x128", "64x64", "128x128x8", "256x128x16", "128x128ea", "128x64ea", "64x64ea", "128x128x8ea", "256x128x16ea", "256x64x8ea"])
@pytest.mark.parametrize("w_il", [False, True], ids=["w-planes", "w-interleaved"])
def test_conv_interleaved_inputs_bit_equal(case, tile, w_il, dev, monkeypatch):
    from fgt_amd import ops
    name, N, H, W, Cin, Cout, k, s, p, d, g = case
    kh, kw = (k, k) if isinstance(k, int) else k
    x = _rand(N, H, W, Cin, seed=1).to(dev)
    w = _rand(Cout, Cin // g, kh, kw, seed=2, scale=1.0 / math.sqrt(Cin // g * kh * kw))
    pc = ops.PackedConv(w.to(dev), _rand(Cout, seed=3).to(dev), groups=g)
    monkeypatch.setattr(ops, "WEIGHTS_INTERLEAVED", False)
    ref = ops.conv2d(x, pc, stride=s, pad=p, dil=d, act="lrelu", tile="128x128", precision="bf16x3")
    monkeypatch.setattr(ops, "WEIGHTS_INTERLEAVED", w_il)
    a = ops.conv2d(x, pc, stride=s, pad=p, dil=d, act="lrelu", tile=tile if not tile.endswith("ea") else "128x128", precision="bf16x3")
    b = ops.conv2d(ops.split(x, interleave=True), pc, stride=s, pad=p, dil=d, act="lrelu", tile=tile, precision="bf16x3")
    c = ops.conv2d(ops.split(x), pc, stride=s, pad=p, dil=d, act="lrelu", tile=tile, precision="bf16x3")
    assert torch.equal(a, ref) and torch.equal(b, ref) and torch.equal(c, ref)


def test_conv_interleaved_two_source_and_out_il(dev):
    from fgt_amd import ops
    N, H, W, g = 2, 15, 27, 2
    x0, o = _rand(N, H, W, 256, seed=1).to(dev), _rand(N, H, W, 384, seed=2).to(dev)
    w, b = _rand(512, 640 // g, 3, 3, seed=3, scale=0.05), _rand(512, seed=4)
    pc = ops.PackedConv(w.to(dev), b.to(dev), groups=g)
    ref = ops.conv2d(x0, pc, x1=o, stride=1, pad=1, act="lrelu", precision="bf16x3")
    for tile in ("128x128x8", "64x64"):            # (tile = auto would route this 3x3 / stride-1 layer to the tap-reusing kernel: tests/test_taps_gpu.py)
        r32, rs = ops.conv2d(ops.split(x0, interleave=True), pc, x1=ops.split(o, interleave=True), stride=1, pad=1, act="lrelu",
                             precision="bf16x3", tile=tile, out_split="both", out_il=True)
        assert torch.equal(r32, ref) and rs.il
        hi, lo = rs.planes()
        want = ops.split(ref)
        assert torch.equal(hi, want.data[0]) and torch.equal(lo, want.data[1])
    # conv -> conv chain in interleaved form
    w2 = _rand(64, 512, 3, 3, seed=5, scale=0.02)
    pc2 = ops.PackedConv(w2.to(dev), None)
    a = ops.conv2d(ref, pc2, pad=1, stride=2, precision="bf16x3")
    bb = ops.conv2d(rs, pc2, pad=1, stride=2, precision="bf16x3")
    assert torch.equal(a, bb)
    with pytest.raises(RuntimeError, match="multiples of 32"):
        ops.conv2d(ops.Split(torch.zeros(1, 8, 8, 80, dtype=torch.bfloat16, device=dev), True),
                   ops.PackedConv(_rand(16, 40, 3, 3, seed=2).to(dev), None), pad=1, precision="bf16x3", tile="64x64")


# ---- wide LDS image (csrc/conv_wide.hip): interleaved inputs + interleaved weights, 8-row x 128-byte LDS-DMA pieces, 8-phase tiles
from fgt_amd import _lib as _fgt_lib

P8_TILES = ["256x256p8w", "256x128p8w"] if "diag" in _fgt_lib.LIB_PATH else []      # the 8-phase schedule exists in diagnostic builds only
WIDE_TILES = ["128x128w", "128x64w", "64x64w", "128x32w", "256x128w", "128x128x8w", "256x128x16w", "256x64x8w", "128x128eaw", "128x64eaw", "64x64eaw",
              "128x128x8eaw", "256x128x16eaw", "256x64x8eaw", "256x128eaw"] + P8_TILES
WIDE_CASES = IL_CASES + [
    ("3x3_cin256_cout384", 1, 20, 36, 256, 384, 3, 1, 1, 1, 1),      # several K-steps per tap, N tile past Npad for the 256-wide tiles
    ("1x1_k768_rows_not_tile_multiple", 1, 1, 1111, 768, 512, 1, 1, 0, 1, 1),
    ("7x7_s3_cin128", 1, 30, 45, 128, 256, 7, 3, 3, 1, 1),
    ("3x3_g2_cin64", 2, 13, 21, 128, 64, 3, 1, 1, 1, 2),
]


@pytest.mark.parametrize("case", WIDE_CASES, ids=[c[0] for c in WIDE_CASES])
@pytest.mark.parametrize("tile", WIDE_TILES)
def test_conv_wide_tiles_bit_equal(case, tile, dev):
    from fgt_amd import ops
    name, N, H, W, Cin, Cout, k, s, p, d, g = case
    kh, kw = (k, k) if isinstance(k, int) else k
    x = _rand(N, H, W, Cin, seed=1).to(dev)
    w = _rand(Cout, Cin // g, kh, kw, seed=2, scale=1.0 / math.sqrt(Cin // g * kh * kw))
    pc = ops.PackedConv(w.to(dev), _rand(Cout, seed=3).to(dev), groups=g)
    ref = ops.conv2d(x, pc, stride=s, pad=p, dil=d, act="lrelu", tile="128x128", precision="bf16x3")
    got = ops.conv2d(ops.split(x, interleave=True), pc, stride=s, pad=p, dil=d, act="lrelu", tile=tile, precision="bf16x3")
    torch.cuda.synchronize()
    assert torch.equal(got, ref), f"{name} tile={tile}: max diff {(got - ref).abs().max().item():.3e}"


def test_conv_wide_two_source_upsample_replicate_out_il(dev):
    """The wave-uniform (tap, source, channel) walk of the wide kernel: group-interleaved two-source concat, nearest-x2 upsample,
    replicate padding, dilation; interleaved split output feeding another wide conv; planes inputs and planes weights are rejected."""
    from fgt_amd import ops
    N, H, W = 2, 15, 27
    x0, o = _rand(N, H, W, 256, seed=1).to(dev), _rand(N, H, W, 384, seed=2).to(dev)
    for g, cout in ((2, 512), (4, 384)):
        w, b = _rand(cout, 640 // g, 3, 3, seed=3, scale=0.05), _rand(cout, seed=4)
        pc = ops.PackedConv(w.to(dev), b.to(dev), groups=g)
        ref = ops.conv2d(x0, pc, x1=o, stride=1, pad=1, act="lrelu", precision="bf16x3")
        for tile in ["128x128x8eaw", "64x64w", "128x128w"] + P8_TILES:
            r32, rs = ops.conv2d(ops.split(x0, interleave=True), pc, x1=ops.split(o, interleave=True), stride=1, pad=1, act="lrelu",
                                 precision="bf16x3", tile=tile, out_split="both", out_il=True)
            assert torch.equal(r32, ref), (g, tile)
            assert torch.equal(rs.data, ops.split(ref, interleave=True).data), (g, tile)
    x = _rand(1, 12, 20, 64, seed=5).to(dev)
    w2, b2 = _rand(48, 64, 3, 3, seed=6, scale=0.1), _rand(48, seed=7)
    pc2 = ops.PackedConv(w2.to(dev), b2.to(dev))
    for kw in (dict(upsample=True, pad=1), dict(pad=2, pad_mode="replicate", dil=2), dict(stride=2, pad=1)):
        ref = ops.conv2d(x, pc2, precision="bf16x3", **kw)
        for tile in ["64x64eaw", "128x64w"] + P8_TILES[1:]:
            assert torch.equal(ops.conv2d(ops.split(x, interleave=True), pc2, precision="bf16x3", tile=tile, **kw), ref), (kw, tile)
    with pytest.raises(RuntimeError, match="not built|unknown tile|wide"):
        ops.conv2d(ops.split(x), pc2, pad=1, precision="bf16x3", tile="128x128w")         # planes input on a wide tile


# ---- fused producers of split tensors: LayerNorm, attention, fold (the GEMM operands of the transformer blocks)
def _same_split(sp, ref32):
    from fgt_amd import ops
    want = ops.split(ref32.contiguous())
    return torch.equal(sp.data[0], want.data[0]) and torch.equal(sp.data[1], want.data[1])


def test_layernorm_split_outputs(dev):
    from fgt_amd import ops
    rows = 777
    x0, x1 = _rand(rows, 512, seed=1, scale=2.0).to(dev), _rand(rows, 256, seed=2).to(dev)
    gA, bA, gB, bB = (_rand(768, seed=s).to(dev) for s in (3, 4, 5, 6))
    a32, b32 = ops.layernorm(x0, gA, bA, x1=x1, gB=gB, bB=bB)
    a, b = ops.layernorm(x0, gA, bA, x1=x1, gB=gB, bB=bB, splitA=True, splitB=True)
    assert _same_split(a, a32) and _same_split(b, b32)
    # mixed: fp32 A, split B written into a row slice of a longer split buffer (the spatial block's key buffer)
    big = ops.Split.empty((rows + 60, 768), dev)
    big.data.zero_()
    a2 = torch.empty(rows, 768, device=dev)
    ops.layernorm(x0, gA, bA, x1=x1, gB=gB, bB=bB, outA=a2, outB=big[:rows])
    assert torch.equal(a2, a32) and _same_split(big[:rows], b32) and float(big.data[:, rows:].float().abs().max()) == 0.0
    g1, b1 = _rand(512, seed=7).to(dev), _rand(512, seed=8).to(dev)
    assert _same_split(ops.layernorm(x0, g1, b1, splitA=True), ops.layernorm(x0, g1, b1))


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_attention_split_output(prec, dev):
    from fgt_amd import ops
    b, t, nh, nw, c = 1, 3, 20, 36, 512
    qkv = _rand(b * t * nh * nw, 3 * c, seed=1).to(dev)
    o32 = ops.attention_temporal(qkv, b, t, nh, nw, 4, 2, c, precision=prec)
    assert _same_split(ops.attention_temporal(qkv, b, t, nh, nw, 4, 2, c, precision=prec, out_split=True), o32)
    bt, h, w, nh, nw = 2, 22, 35, 24, 40
    q, k, v = (_rand(bt * nh * nw, c, seed=s).to(dev) for s in (2, 3, 4))
    kg, vg = _rand(bt * 60, c, seed=5).to(dev), _rand(bt * 60, c, seed=6).to(dev)
    o32 = ops.attention_spatial(q, k, v, kg, vg, bt, h, w, nh, nw, 4, 8, 60, precision=prec)
    assert _same_split(ops.attention_spatial(q, k, v, kg, vg, bt, h, w, nh, nw, 4, 8, 60, precision=prec, out_split=True), o32)


def test_fold_relu_split_output(dev):
    from fgt_amd import ops
    frames, th, tw, Cc, k, s, p, Hf, Wf = 3, 20, 36, 40, 7, 3, 3, 60, 108
    Y = _rand(frames * th * tw, k * k * Cc, seed=1).to(dev)
    f32 = ops.fold(Y, frames, th, tw, Cc, k, s, p, Hf, Wf, normalize=True)
    fr = ops.fold(Y, frames, th, tw, Cc, k, s, p, Hf, Wf, normalize=True, relu=True)
    assert torch.equal(fr, f32.clamp_min(0))
    assert _same_split(ops.fold(Y, frames, th, tw, Cc, k, s, p, Hf, Wf, normalize=True, relu=True, out_split=True), fr)
    # ReLU-then-split feeding the 7x7/s3 conv == the conv with ReLU on the gathered fp32 values (the two FFN formulations)
    w = _rand(512, Cc, k, k, seed=2, scale=0.05)
    pc = ops.PackedConv(w.to(dev), None)
    a = ops.conv2d(f32, pc, stride=s, pad=p, in_relu=True, precision="bf16x3")
    bsp = ops.conv2d(ops.fold(Y, frames, th, tw, Cc, k, s, p, Hf, Wf, normalize=True, relu=True, out_split=True), pc, stride=s, pad=p,
                     precision="bf16x3")
    assert torch.equal(a, bsp)


# ---- interleaved split outputs of the fused producers (ps = 32) and the model on them
def test_layernorm_and_fold_interleaved_outputs(dev, monkeypatch):
    from fgt_amd import ops
    rows = 501
    x0, x1 = _rand(rows, 512, seed=1, scale=2.0).to(dev), _rand(rows, 256, seed=2).to(dev)
    gA, bA, gB, bB = (_rand(768, seed=s).to(dev) for s in (3, 4, 5, 6))
    pa, pb = ops.layernorm(x0, gA, bA, x1=x1, gB=gB, bB=bB, splitA=True, splitB=True)             # planes (default mode: fp32)
    ia, ib = ops.Split.empty((rows + 7, 768), dev, interleaved=True), ops.Split.empty((rows, 768), dev, interleaved=True)
    ia.data.zero_()
    ops.layernorm(x0, gA, bA, x1=x1, gB=gB, bB=bB, outA=ia[:rows], outB=ib)                          # a row slice of a longer interleaved buffer
    for il, pl in ((ia[:rows], pa), (ib, pb)):
        hi, lo = il.planes()
        assert torch.equal(hi, pl.data[0]) and torch.equal(lo, pl.data[1])
    assert float(ia.data[rows:].float().abs().max()) == 0.0
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", "bf16x3")                                   # split_il(): new split tensors are interleaved
    auto = ops.layernorm(x0, gA, bA, x1=x1, splitA=True)
    assert auto.il and torch.equal(auto.data, ia[:rows].data)
    frames, th, tw, Cc, k, s, p, Hf, Wf = 2, 20, 36, 128, 7, 3, 3, 60, 108
    Y = _rand(frames * th * tw, k * k * Cc, seed=7).to(dev)
    res = _rand(frames, Hf, Wf, Cc, seed=8).to(dev)
    f_il = ops.fold(Y, frames, th, tw, Cc, k, s, p, Hf, Wf, normalize=False, res=res, out_split=True)
    monkeypatch.setattr(ops, "SPLIT_INTERLEAVED", False)
    f_pl = ops.fold(Y, frames, th, tw, Cc, k, s, p, Hf, Wf, normalize=False, res=res, out_split=True)
    assert f_il.il and not f_pl.il
    hi, lo = f_il.planes()
    assert torch.equal(hi, f_pl.data[0]) and torch.equal(lo, f_pl.data[1])


def test_fgt_forward_bit_equal_with_interleaved_split_tensors(dev, monkeypatch):
    """The bf16x3 model with its GEMM operands in the interleaved layout (LayerNorm / fold / conv epilogue write [hi 32 | lo 32] rows, the
    consumers may run on the wide kernel) == the same model on planes, bit for bit, whatever tiles the autotuner picks."""
    from fgt_amd import ops
    from fgt_amd.fgt_model import DEFAULT_CONFIG, Model
    from fgt_amd.synth import synth_state_dict
    from util import fgt_inputs
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", "bf16x3")
    monkeypatch.setattr(ops, "DEFAULT_ATTN_PRECISION", "bf16x3")
    m = Model(dict(DEFAULT_CONFIG)).eval()
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0), strict=True)
    m = m.to(dev)
    mf, fl, ms = (t.to(dev) for t in fgt_inputs(64, 96, 3, seed=4))
    outs = {}
    for il in (True, False):
        monkeypatch.setattr(ops, "SPLIT_INTERLEAVED", il)
        outs[il] = m(mf, fl, ms).clone()
    assert torch.equal(outs[True], outs[False])
    # and with every conv forced onto a wide tile where its input is interleaved (the autotuner may not have picked one above)
    real = ops.conv2d

    def forced(x, pc, *a, **kw):
        # (3x3 / stride-1 layers, with or without nearest upsampling, are routed to the tap-reusing kernel by geometry in both layouts: not forced)
        tap_routed = pc.kw >= 3 and kw.get("stride", 1) == 1 and kw.get("pad_mode", "zeros") == "zeros"
        # (the 2x2 sub-pixel form of an upsampled layer picks among the tiles whose N width divides its ps_c: not forced either)
        if isinstance(x, ops.Split) and x.il and kw.get("tile") is None and pc.Cout // pc.groups > 4 and not tap_routed and not kw.get("_phase_pad"):
            kw["tile"] = "128x128x8eaw"
        return real(x, pc, *a, **kw)
    monkeypatch.setattr(ops, "SPLIT_INTERLEAVED", True)
    monkeypatch.setattr(ops, "conv2d", forced)
    assert torch.equal(m(mf, fl, ms), outs[False])
