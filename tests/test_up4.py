"""The 2x2 sub-pixel form of "nearest x2 upsampling + 3x3 convolution" (fgt_conv_desc.ps_phase_pad, ABI 9; ops.up4_weights).
CPU part: the weight transform against the reference formulation (network_blocks_2d.py:46-60) in fp64; GPU part: the route against fp64 and
against the upsampled 3x3 form on the tap kernels."""
import math

import pytest
import torch
import torch.nn.functional as F


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


@pytest.mark.parametrize("shape", [(8, 5, 6, 7), (4, 3, 1, 1), (6, 2, 5, 1), (3, 4, 2, 9)])
def test_up4_weights_reproduce_upsample_then_conv(shape):
    from fgt_amd import ops
    Cout, Cin, H, W = shape
    w, x = _rand(Cout, Cin, 3, 3, seed=1), _rand(2, Cin, H, W, seed=2).double()
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w.double(), padding=1)
    wp = ops.up4_weights(w, Cout + 3).view(2, 2, Cout + 3, Cin, 2, 2)                      # padded sub-pixel blocks: zero rows behind the real ones
    assert float(wp[:, :, Cout:].abs().max()) == 0.0 and torch.equal(wp[:, :, :Cout].reshape(-1), ops.up4_weights(w).reshape(-1))
    w4 = ops.up4_weights(w).double().view(2, 2, Cout, Cin, 2, 2)
    out = torch.zeros_like(ref)
    for a in range(2):
        for b in range(2):
            xp = F.pad(x, (1 - b, b, 1 - a, a))          # sub-pixel (a, b): padding (1 - a, 1 - b) in front, (a, b) behind
            out[:, :, a::2, b::2] = F.conv2d(xp, w4[a, b])
    assert (out - ref).abs().max().item() <= 4e-7 * ref.abs().max().item() + 1e-7      # (the sums are rounded to fp32 once)


def _ref64(xs, x1s, w, b, dev):
    xd = xs.float().double() if x1s is None else torch.cat([xs.float().double(), x1s.float().double()], -1)
    up = F.interpolate(xd.permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
    return F.conv2d(up, w.to(dev).double(), None if b is None else b.to(dev).double(), 1, 1).permute(0, 2, 3, 1)


CASES = [(3, 9, 13, 64, 0, 64), (2, 15, 27, 96, 96, 48), (1, 30, 54, 128, 0, 128), (2, 7, 10, 64, 0, 192), (1, 1, 1, 32, 0, 64), (2, 33, 5, 64, 32, 64), (2, 8, 11, 192, 192, 96)]


@pytest.mark.gpu
@pytest.mark.parametrize("il", [False, True], ids=["planes", "interleaved"])
def test_up4_route_vs_fp64_and_the_upsampled_form(il, dev):
    """tile = auto on an eligible layer takes the 2x2 sub-pixel form; an explicit tile keeps the upsampled 3x3 form: both against fp64 of the same
    operands (exact weights: the route rounds weight SUMS, so it is compared with the fp32 weights' fp64 result), odd sizes, two sources, 1x1 maps."""
    from fgt_amd import ops
    assert ops.UP4
    for (N, H, W, C0, C1, Cout) in CASES:
        x, x1 = _rand(N, H, W, C0, seed=1).to(dev), (_rand(N, H, W, C1, seed=2).to(dev) if C1 else None)
        w, b = _rand(Cout, C0 + C1, 3, 3, seed=3, scale=1.0 / math.sqrt(9 * (C0 + C1))), _rand(Cout, seed=4)
        pc = ops.PackedConv(w.to(dev), b.to(dev))
        xs, x1s = ops.split(x, interleave=il), (ops.split(x1, interleave=il) if C1 else None)
        ref = F.leaky_relu(_ref64(xs, x1s, w, b, dev), 0.2)
        scale = ref.abs().max().item()
        old = ops.conv2d(xs, pc, x1=x1s, pad=1, upsample=True, act="lrelu", tile="128x128", precision="bf16x3")
        got = ops.conv2d(xs, pc, x1=x1s, pad=1, upsample=True, act="lrelu", precision="bf16x3")
        torch.cuda.synchronize()
        e_old, e = (old.double() - ref).abs().max().item(), (got.double() - ref).abs().max().item()
        # (the reference holds the exact fp32 weights: both forms carry the 2^-17 rounding of their hi + lo weight images, 3e-6 of the scale over K = 9 x 64...384)
        assert got.shape == (N, 2 * H, 2 * W, Cout) and e <= max(1.5 * e_old, 6e-6 * scale), f"{(N, H, W, C0, C1, Cout)}: {e:.3e} (upsampled form {e_old:.3e}, scale {scale:.2e})"
        assert (got - old).abs().max().item() <= 6e-6 * scale
        assert not torch.equal(got, old) or H * W == 1           # (another summation: the route really ran)
        # both output forms, interleaved when the channel count allows
        o32, os_ = ops.conv2d(xs, pc, x1=x1s, pad=1, upsample=True, act="lrelu", precision="bf16x3", out_split="both", out_il=il and Cout % 32 == 0, out_h=False)
        assert torch.equal(o32, got) and (os_.float() - o32).abs().max().item() <= 2.0 ** -15 * scale
        only = ops.conv2d(xs, pc, x1=x1s, pad=1, upsample=True, act="lrelu", precision="bf16x3", out_split="only", out_il=il and Cout % 32 == 0, out_h=False)
        assert torch.equal(only.data, os_.data)
        print(f"[parity] up4 {N}x{H}x{W} {C0}+{C1}->{Cout} ({'interleaved' if il else 'planes'}): max |2x2 form - fp64| {e:.2e}, |upsampled 3x3 - fp64| {e_old:.2e} (scale {scale:.2e})")


@pytest.mark.gpu
def test_up4_every_tile_is_bit_identical_and_epilogues(dev):
    """The explicit form (ops.conv2d(..., ps=..., _phase_pad=64)) on every conv_split / conv_wide tile that accepts it: one result; tiles whose N
    width does not divide ps_c decline; the gated block's `mul` (aux1 shaped like the OUTPUT) and `add` + second activation."""
    from fgt_amd import ops
    N, H, W, C, Cout = 2, 12, 20, 64, 64
    w, b = _rand(Cout, C, 3, 3, seed=3, scale=1.0 / math.sqrt(9 * C)), _rand(Cout, seed=4)
    pc = ops.PackedConv(w.to(dev), b.to(dev))
    for il in (False, True):
        xs = ops.split(_rand(N, H, W, C, seed=1).to(dev), interleave=il)
        first, ran = None, []
        for t in ops.TILE_CANDIDATES:
            try:
                got = ops.conv2d(xs, ops._up4_pack(pc), stride=1, pad=1, act="lrelu", ps=(2, Cout, 2 * Cout, 2 * H, 2 * W), _phase_pad=64, tile=t, precision="bf16x3")
            except RuntimeError:
                continue
            torch.cuda.synchronize()
            ran.append(t)
            first = got if first is None else first
            assert torch.equal(got, first), t
        assert len(ran) >= 3 and all("128x128" not in t and "256x128" not in t for t in ran), ran           # N width 64 only (ps_c = 64)
        assert torch.equal(ops.conv2d(xs, pc, pad=1, upsample=True, act="lrelu", precision="bf16x3"), first)
        # gated block: sigmoid gate, then feature conv times the gate (network_blocks_2d.py:86-91)
        wg = _rand(Cout, C, 3, 3, seed=5, scale=1.0 / math.sqrt(9 * C))
        pg = ops.PackedConv(wg.to(dev), None)
        gate = ops.conv2d(xs, pg, pad=1, upsample=True, act="sigmoid", precision="bf16x3")
        y = ops.conv2d(xs, pc, pad=1, upsample=True, act="lrelu", epi="mul", aux1=gate, precision="bf16x3")
        ref = F.leaky_relu(_ref64(xs, None, w, b, dev), 0.2) * torch.sigmoid(_ref64(xs, None, wg, None, dev))
        assert (y.double() - ref).abs().max().item() <= 6e-6 * ref.abs().max().item()
        res = _rand(N, 2 * H, 2 * W, Cout, seed=6).to(dev)
        y = ops.conv2d(xs, pc, pad=1, upsample=True, act=None, epi="add", aux1=res, act2="relu", precision="bf16x3")
        ref = F.relu(_ref64(xs, None, w, b, dev) + res.double())
        assert (y.double() - ref).abs().max().item() <= 6e-6 * ref.abs().max().item()


@pytest.mark.gpu
def test_up4_rejects_what_it_does_not_serve(dev):
    from fgt_amd import ops
    xs = ops.split(_rand(1, 8, 8, 64, seed=1).to(dev))
    pc = ops.PackedConv(_rand(64, 64, 3, 3, seed=2, scale=0.05).to(dev), None)
    q = ops._up4_pack(pc)
    with pytest.raises(RuntimeError):           # a tap tile
        ops.conv2d(xs, q, stride=1, pad=1, ps=(2, 64, 128, 16, 16), _phase_pad=64, tile="128x64t", precision="bf16x3")
    with pytest.raises(RuntimeError):           # N width 128 on ps_c = 64
        ops.conv2d(xs, q, stride=1, pad=1, ps=(2, 64, 128, 16, 16), _phase_pad=64, tile="128x128", precision="bf16x3")
    # layers the route leaves alone: fewer than 32 output channels, replicate padding, an explicit tile
    pc2 = ops.PackedConv(_rand(24, 64, 3, 3, seed=2, scale=0.05).to(dev), None)
    a = ops.conv2d(xs, pc2, pad=1, upsample=True, precision="bf16x3")
    assert torch.equal(a, ops.conv2d(xs, pc2, pad=1, upsample=True, tile="128x64t", precision="bf16x3"))
    assert torch.equal(ops.conv2d(xs, pc, pad=1, upsample=True, pad_mode="replicate", precision="bf16x3"), ops.conv2d(xs, pc, pad=1, upsample=True, pad_mode="replicate", tile="128x128", precision="bf16x3"))


@pytest.mark.gpu
def test_up4_declined_layer_falls_back_to_the_upsampled_form(dev, monkeypatch):
    """A layer the 2x2 form's argument validation declines (here: a weight image whose sub-pixel blocks are not padded to the tile width) runs in
    the upsampled 3x3 form, with a warning, and is remembered."""
    from fgt_amd import ops
    xs = ops.split(_rand(1, 8, 8, 64, seed=1).to(dev))
    pc = ops.PackedConv(_rand(48, 64, 3, 3, seed=2, scale=0.05).to(dev), None)
    want = ops.conv2d(xs, pc, pad=1, upsample=True, precision="bf16x3", tile="128x64t")

    def bad_pack(p):
        q = ops.PackedConv(ops.up4_weights(p.w[0, :p.Cout, :p.K].reshape(p.Cout, 3, 3, p.Cg).permute(0, 3, 1, 2)), None, pad_cin_to4=False)
        q.up4_c = p.Cout                       # 48: not a multiple of 64
        return q
    monkeypatch.setattr(ops, "_up4_pack", bad_pack)
    with pytest.warns(UserWarning, match="2x2 sub-pixel form declined"):
        got = ops.conv2d(xs, pc, pad=1, upsample=True, precision="bf16x3")
    assert torch.equal(got, want) and "_up4_declined" in pc.__dict__
    assert torch.equal(ops.conv2d(xs, pc, pad=1, upsample=True, precision="bf16x3"), want)      # remembered: no second attempt
