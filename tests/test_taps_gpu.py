"""The tap-reusing conv kernel (csrc/conv_taps.hip, tile codes + 200) on a real MI355X.

It computes the same products as the other bf16x3 kernels but accumulates them in the order (ky, chunk, kx) instead of (ky, kx, chunk):
NOT bit-identical to them, so its gate is (i) distance to fp64 on the same split operands no larger than the other kernels', (ii) bit
equality among its own tiles, (iii) the routing: a layer's kernel family is decided by geometry, never by tuning."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
from fgt_amd import _lib

TAPS = ["128x128x8t", "128x128t", "128x64t", "128x64x8t", "64x64t", "128x128it", "256x128it", "256x256it"]      # ...it: conv_taps_il.hip (requests interleaved into the matrix work)
PP_ONLY_KX = ("128x128it", "256x128it", "256x256it")            # the interleaved tiles serve the kx-reuse geometry only: no k x 1 (transposed) layers, no upsampling
if "diag" in _lib.LIB_PATH:        # diagnostic builds: the same kernel with register-fed weights (csrc/diag/conv_taps_breg.hip), bit-identical
    TAPS += ["128x128x8r", "128x128r", "128x64r", "64x64r"]


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


CASES = [
    # name, N, H, W, C0, C1, Cout, (kh, kw), pad, dil, groups
    ("3x3_64_128", 2, 20, 28, 64, 0, 128, (3, 3), (1, 1), 1, 1),
    ("3x3_rows_cross_images", 3, 7, 9, 32, 0, 64, (3, 3), (1, 1), 1, 1),            # W = 9: a 128-row tile spans 14 image rows and 2-3 images
    ("3x3_w_smaller_than_halo", 2, 33, 5, 32, 0, 32, (3, 3), (1, 1), 1, 1),
    ("1x5_gru", 2, 15, 27, 128, 256, 128, (1, 5), (0, 2), 1, 1),                    # RAFT SepConvGRU horizontal pass, two sources
    ("5x1_gru_vertical", 2, 15, 27, 128, 256, 128, (5, 1), (2, 0), 1, 1),           # RAFT SepConvGRU vertical pass: ky taps reused, tile rows in (n, x, y) order
    ("3x1_h_smaller_than_halo", 2, 5, 33, 32, 0, 32, (3, 1), (1, 0), 1, 1),
    ("7x1_dil2_cout_200", 1, 40, 21, 64, 0, 200, (7, 1), (6, 0), 2, 1),
    ("5x1_cout_50_general_epilogue", 2, 20, 19, 64, 0, 50, (5, 1), (2, 0), 1, 1),   # Cout % 4 != 0: workgroup-wide epilogue, rows mapped back from (n, x, y)
    ("3x3_cout_50_general_epilogue", 2, 11, 23, 32, 32, 50, (3, 3), (1, 1), 1, 1),
    ("1x1_not_served", 1, 15, 27, 64, 0, 64, (1, 1), (0, 0), 1, 1),                 # no taps to reuse: the other kernels
    ("3x3_dil2", 1, 30, 27, 96, 0, 96, (3, 3), (2, 2), 2, 1),
    ("3x3_dil8", 1, 40, 44, 192, 0, 192, (3, 3), (8, 8), 8, 1),                     # LAFC middle: (kw - 1) * dw = 16 = the halo
    ("3x3_g2_two_source", 2, 15, 27, 256, 384, 512, (3, 3), (1, 1), 1, 2),          # encoder group-interleaved concat
    ("7x7", 1, 21, 25, 32, 0, 48, (7, 7), (3, 3), 1, 1),
    ("3x3_cout_not_tile_multiple", 1, 17, 23, 64, 0, 200, (3, 3), (1, 1), 1, 1),
]


def _ref64(xs, x1s, w, b, pad, dil, g):
    """fp64 convolution of the values the split operands stand for (hi + lo), with the weights' hi + lo as the kernel sees them."""
    from fgt_amd import ops
    x = xs.float().double()
    if x1s is not None:
        N, H, W, C0 = x.shape
        y = x1s.float().double()
        x = torch.cat([x.view(N, H, W, g, C0 // g), y.view(N, H, W, g, y.shape[-1] // g)], -1).reshape(N, H, W, -1)
    wh = w.to(torch.bfloat16)
    wd = (wh.float() + (w - wh.float()).to(torch.bfloat16).float()).double()
    out = F.conv2d(x.permute(0, 3, 1, 2), wd, b.double(), 1, pad, dil, g)
    return F.leaky_relu(out, 0.2).permute(0, 2, 3, 1)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("il", [False, True], ids=["planes", "interleaved"])
def test_taps_kernel_vs_fp64_and_other_kernels(case, il, dev):
    from fgt_amd import ops
    name, N, H, W, C0, C1, Cout, (kh, kw), pad, dil, g = case
    x = _rand(N, H, W, C0, seed=1).to(dev)
    x1 = _rand(N, H, W, C1, seed=2).to(dev) if C1 else None
    Cg = (C0 + C1) // g
    w = _rand(Cout, Cg, kh, kw, seed=3, scale=1.0 / math.sqrt(Cg * kh * kw))
    b = _rand(Cout, seed=4)
    pc = ops.PackedConv(w.to(dev), b.to(dev), groups=g)
    xs, x1s = ops.split(x, interleave=il), (ops.split(x1, interleave=il) if C1 else None)
    ref = _ref64(xs, x1s, w.to(dev), b.to(dev), pad, dil, g)
    other = ops.conv2d(xs, pc, x1=x1s, pad=pad, dil=dil, act="lrelu", tile="128x128", precision="bf16x3")       # explicit tile: conv_split / conv_wide family
    assert torch.equal(other, ops.conv2d(x, pc, x1=x1, pad=pad, dil=dil, act="lrelu", tile="128x128", precision="bf16x3"))
    e_other = (other.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    auto = ops.conv2d(xs, pc, x1=x1s, pad=pad, dil=dil, act="lrelu", precision="bf16x3")                           # routed by geometry
    if kw == 1 and kh == 1:
        assert torch.equal(auto, other)
        with pytest.raises(RuntimeError, match="does not serve"):
            ops.conv2d(xs, pc, x1=x1s, pad=pad, dil=dil, act="lrelu", tile="128x128x8t", precision="bf16x3")
        return
    first = None
    for t in TAPS:
        if kw == 1 and t.endswith("r"):
            continue                         # (the register-fed-weights experiment has no transposed mode)
        if (kw == 1 and t in PP_ONLY_KX) or (t == "256x256it" and kw != 3):
            with pytest.raises(RuntimeError, match="not serve"):
                ops.conv2d(xs, pc, x1=x1s, pad=pad, dil=dil, act="lrelu", tile=t, precision="bf16x3")
            continue
        got = ops.conv2d(xs, pc, x1=x1s, pad=pad, dil=dil, act="lrelu", tile=t, precision="bf16x3")
        torch.cuda.synchronize()
        e = (got.double() - ref).abs().max().item()
        assert e <= max(2.0 * e_other, 2e-6 * scale) and e <= 2e-5 * scale, f"{name} {t}: {e:.3e} vs other kernels {e_other:.3e} (scale {scale:.2e})"
        first = got if first is None else first
        assert torch.equal(got, first), f"{name}: tile {t} differs from {TAPS[0]}"
    routed = kw > 1 or H >= 16             # (k x 1 over a short axis: the kernel serves it, the library does not prefer it)
    assert torch.equal(auto, first if routed else other), "tile = auto: the tap-reusing kernel exactly on the layers the library routes to it"
    print(f"[parity] conv_taps {name} ({'interleaved' if il else 'planes'}): max |taps - fp64| {(first.double() - ref).abs().max().item():.2e}, "
          f"|conv_split - fp64| {e_other:.2e}, |taps - conv_split| {(first - other).abs().max().item():.2e} (outputs up to {scale:.2f})")


def test_taps_epilogues_and_split_outputs(dev):
    """GRU / mul / add epilogues, fp32 + split outputs, output written into a channel slice — through the tap kernel (RAFT's update block:
    horizontal 1x5 pass, and the vertical 5x1 pass whose tile rows are in (n, x, y) order: aux operands and outputs are mapped back)."""
    from fgt_amd import ops
    B, H, W = 2, 17, 27                     # (H >= 16: the vertical pass is routed to the tap-reusing kernel as well)
    rows = B * H * W
    net, xb = _rand(B, H, W, 128, seed=1).to(dev), _rand(B, H, W, 256, seed=2).to(dev)
    z, hprev = torch.sigmoid(_rand(rows, 128, seed=3)).to(dev), _rand(rows, 128, seed=4).to(dev)
    ns, xs = ops.split(net), ops.split(xb)
    for ksz, pad in (((1, 5), (0, 2)), ((5, 1), (2, 0))):
        w, b = _rand(128, 384, *ksz, seed=5, scale=0.03), _rand(128, seed=6)
        pc = ops.PackedConv(w.to(dev), b.to(dev))
        for epi, kw in (("gru", dict(act="tanh", epi="gru", aux1=z, aux2=hprev)), ("mul", dict(act="sigmoid", epi="mul", aux1=hprev)), ("none", dict(act="sigmoid"))):
            ref = ops.conv2d(ns, pc, x1=xs, pad=pad, tile="128x128", precision="bf16x3", **kw)
            o32, osp = ops.conv2d(ns, pc, x1=xs, pad=pad, precision="bf16x3", out_split="both", **kw)
            assert (o32 - ref).abs().max().item() <= 3e-6 * max(1.0, ref.abs().max().item()), (ksz, epi)
            assert torch.equal(osp.data, ops.split(o32).data)
            only = ops.conv2d(ns, pc, x1=xs, pad=pad, precision="bf16x3", out_split="only", **kw)
            assert torch.equal(only.data, osp.data)
            nchw = ops.conv2d(ns, pc, x1=xs, pad=pad, precision="bf16x3", out_nchw=True, tile="128x64t", **kw)      # general (workgroup-wide) epilogue
            # (the two epilogue code paths may contract their multiply-adds differently: last-bit agreement, not bit equality)
            assert (nchw.permute(0, 2, 3, 1) - o32.view(B, H, W, 128)).abs().max().item() <= 1e-6 * max(1.0, o32.abs().max().item()), (ksz, epi)
    wide = ops.Split.empty((rows, 256), dev)
    wide.data.zero_()
    w2 = _rand(128, 256, 3, 3, seed=7, scale=0.02)
    pc2 = ops.PackedConv(w2.to(dev), None)
    full = ops.conv2d(xs, pc2, pad=1, act="relu", precision="bf16x3", out_split="only")
    ops.conv2d(xs, pc2, pad=1, act="relu", precision="bf16x3", out_split="only", out_s=wide.channels(128, 256))
    assert torch.equal(wide.data[:, :, 128:], full.data.view(2, rows, 128)) and float(wide.data[:, :, :128].float().abs().max()) == 0.0


@pytest.mark.parametrize("il", [False, True], ids=["planes", "interleaved"])
def test_taps_nearest_upsampling(il, dev):
    """desc.upsample (nearest x2 ahead of the conv: FGT's decoder, LAFC's decoder): the tile walks the output grid, LDS rows fetch input
    pixel ((y' + dy) >> 1, x' >> 1).  Two sources (LAFC concatenates the skip), odd sizes, an image boundary inside a tile."""
    from fgt_amd import ops
    for (N, H, W, C0, C1, Cout) in ((3, 9, 13, 64, 0, 64), (2, 15, 27, 96, 96, 48), (1, 30, 54, 128, 0, 200), (2, 7, 10, 32, 0, 50)):
        x = _rand(N, H, W, C0, seed=1).to(dev)
        x1 = _rand(N, H, W, C1, seed=2).to(dev) if C1 else None
        w, b = _rand(Cout, C0 + C1, 3, 3, seed=3, scale=1.0 / math.sqrt(9 * (C0 + C1))), _rand(Cout, seed=4)
        pc = ops.PackedConv(w.to(dev), b.to(dev))
        xs, x1s = ops.split(x, interleave=il), (ops.split(x1, interleave=il) if C1 else None)
        xd = xs.float().double() if C1 == 0 else torch.cat([xs.float().double(), x1s.float().double()], -1)
        up = F.interpolate(xd.permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
        wh = w.to(dev).to(torch.bfloat16)
        wd = (wh.float() + (w.to(dev) - wh.float()).to(torch.bfloat16).float()).double()
        ref = F.leaky_relu(F.conv2d(up, wd, b.to(dev).double(), 1, 1), 0.2).permute(0, 2, 3, 1)
        other = ops.conv2d(xs, pc, x1=x1s, pad=1, upsample=True, act="lrelu", tile="128x128", precision="bf16x3")
        e_other, scale = (other.double() - ref).abs().max().item(), ref.abs().max().item()
        first = None
        for t in [t for t in TAPS if not t.endswith("r") and t not in PP_ONLY_KX]:
            got = ops.conv2d(xs, pc, x1=x1s, pad=1, upsample=True, act="lrelu", tile=t, precision="bf16x3")
            torch.cuda.synchronize()
            e = (got.double() - ref).abs().max().item()
            assert e <= max(2.0 * e_other, 2e-6 * scale) and e <= 2e-5 * scale, f"{t}: {e:.3e} vs other kernels {e_other:.3e} (scale {scale:.2e})"
            first = got if first is None else first
            assert torch.equal(got, first), t
        auto = ops.conv2d(xs, pc, x1=x1s, pad=1, upsample=True, act="lrelu", precision="bf16x3")
        if ops.UP4 and Cout % 4 == 0 and Cout >= 32 and (C0 + C1) % 32 == 0:      # tile = auto: the 2x2 sub-pixel form (tests/test_up4.py), another summation
            assert (auto.double() - ref).abs().max().item() <= 4e-6 * scale
        else:                                                        # tile = auto: routed to the tap kernel
            assert torch.equal(auto, first)
        print(f"[parity] conv_taps upsample x2 {N}x{H}x{W} {C0}+{C1}->{Cout} ({'interleaved' if il else 'planes'}): max |taps - fp64| {e:.2e}, |conv_split - fp64| {e_other:.2e}")


def test_taps_routing_is_geometry_only(dev):
    from fgt_amd import ops
    xf = _rand(1, 16, 20, 64, seed=1).to(dev)
    x = ops.split(xf)
    pc = ops.PackedConv(_rand(64, 64, 3, 3, seed=2, scale=0.05).to(dev), None)
    base = ops.conv2d(x, pc, pad=1, precision="bf16x3")                               # tile = auto on an eligible layer: the tap kernel
    assert torch.equal(base, ops.conv2d(x, pc, pad=1, precision="bf16x3", tile="128x128x8t"))
    # not served: stride 2, replicate padding, "valid" padding, Cin / groups not a multiple of 32, fp32 inputs — all run on the
    # other kernels, bit-identical to the register-staged bf16x3 kernel on fp32 inputs
    for kw in (dict(stride=2, pad=1), dict(pad=1, pad_mode="replicate"), dict(pad=0)):
        a = ops.conv2d(x, pc, precision="bf16x3", **kw)
        assert torch.equal(a, ops.conv2d(xf, pc, precision="bf16x3", tile="128x128", **kw)), kw
    x40 = _rand(1, 16, 20, 40, seed=3).to(dev)
    pc40 = ops.PackedConv(_rand(64, 40, 3, 3, seed=4, scale=0.05).to(dev), None)
    assert torch.equal(ops.conv2d(ops.split(x40), pc40, pad=1, precision="bf16x3"), ops.conv2d(x40, pc40, pad=1, precision="bf16x3", tile="128x128"))
    # an explicit tile of another family on an eligible layer selects that family (A/B measurements): equal to the fp32-input kernel
    assert torch.equal(ops.conv2d(x, pc, pad=1, precision="bf16x3", tile="128x128x8ea"), ops.conv2d(xf, pc, pad=1, precision="bf16x3", tile="128x128"))


def test_bias_map_in_front_of_the_activation(dev):
    """fgt_conv_desc.ld_bias (ABI 7): the bias as an [M, Cout] map — RAFT's SepConvGRU convs take the convolution of the iteration-invariant
    context features this way (RAFT/update.py:45-58).  Every epilogue flavour (none / mul / GRU), fp32 + split outputs, the horizontal pass and the
    vertical one (tile rows in (n, x, y) order: the map is read at the mapped-back pixel), the register-staged fp32 kernel, the general epilogue."""
    from fgt_amd import ops
    B, H, W = 2, 17, 27
    rows = B * H * W
    net, xb = _rand(B, H, W, 128, seed=1).to(dev), _rand(B, H, W, 128, seed=2).to(dev)
    z, hprev = torch.sigmoid(_rand(rows, 128, seed=3)).to(dev), _rand(rows, 128, seed=4).to(dev)
    bmap = _rand(B, H, W, 128, seed=8).to(dev)
    ns, xs = ops.split(net), ops.split(xb)
    for ksz, pad in (((1, 5), (0, 2)), ((5, 1), (2, 0))):
        pc = ops.PackedConv(_rand(128, 256, *ksz, seed=5, scale=0.03).to(dev), None)
        lin = ops.conv2d(ns, pc, x1=xs, pad=pad, precision="bf16x3").view(rows, 128) + bmap.view(rows, 128)      # the pre-activation value
        want = {"none": torch.sigmoid(lin), "mul": torch.sigmoid(lin) * hprev, "gru": (1 - z) * hprev + z * torch.tanh(lin)}
        for epi, kw in (("gru", dict(act="tanh", epi="gru", aux1=z, aux2=hprev)), ("mul", dict(act="sigmoid", epi="mul", aux1=hprev)), ("none", dict(act="sigmoid"))):
            o32, osp = ops.conv2d(ns, pc, x1=xs, pad=pad, precision="bf16x3", bias_map=bmap, out_split="both", **kw)
            assert (o32.view(rows, 128) - want[epi]).abs().max().item() <= 2e-6, (ksz, epi)
            assert torch.equal(osp.data, ops.split(o32).data)
            for t in ("128x128", "128x64t", "128x128x8t"):
                assert torch.equal(ops.conv2d(ns, pc, x1=xs, pad=pad, precision="bf16x3", bias_map=bmap, tile=t, **kw), o32) or t == "128x128", (ksz, epi, t)
            f32 = ops.conv2d(net, pc, x1=xb, pad=pad, precision="fp32", bias_map=bmap, **kw)                         # register-staged exact kernel
            assert (f32.view(rows, 128) - want[epi]).abs().max().item() <= 2e-5, (ksz, epi)
            nchw = ops.conv2d(ns, pc, x1=xs, pad=pad, precision="bf16x3", bias_map=bmap, out_nchw=True, tile="128x64t", **kw)      # general epilogue
            assert (nchw.permute(0, 2, 3, 1) - o32).abs().max().item() <= 1e-6, (ksz, epi)
