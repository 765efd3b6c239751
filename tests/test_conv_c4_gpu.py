"""The 4-channel-input convolution kernel (csrc/conv_c4.hip) on a real MI355X: the first conv of FGT's frame encoder (3x3 stride 2, RGB + mask),
of the flow encoders and LAFC (5x5 on the flow padded to 4 channels, replicate padding) and of RAFT's motion encoder (7x7).  It gathers the fp32
input straight into MFMA fragments; same products as the register-staged bf16x3 kernel, another fp32 summation order — so its gate is the
distance to an fp64 convolution of the split operand values, no larger than the other kernel's, and the routing (geometry only)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _hl(t):
    hi = t.to(torch.bfloat16)
    return (hi.float() + (t - hi.float()).to(torch.bfloat16).float()).double()


CASES = [
    # name, N, H, W, Cout, k, stride, pad, pad_mode, dil, in_relu
    ("frame_enc_3x3_s2", 3, 24, 40, 64, 3, 2, 1, "zeros", 1, False),
    ("flow_enc_5x5_replicate", 2, 24, 40, 64, 5, 1, 2, "replicate", 1, False),
    ("raft_convf1_7x7_c128", 2, 15, 27, 128, 7, 1, 3, "zeros", 1, False),
    ("lafc_5x5_c48", 2, 17, 23, 48, 5, 1, 2, "zeros", 1, False),
    ("odd_rows_cross_images_3x3", 5, 7, 9, 64, 3, 1, 1, "zeros", 1, True),
    ("5x5_dil2_replicate", 1, 20, 31, 64, 5, 1, 4, "replicate", 2, False),
    ("bench_shape_5x5", 2, 240, 432, 64, 5, 1, 2, "replicate", 1, False),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_c4_vs_fp64_and_register_staged_kernel(case, dev):
    from fgt_amd import _lib, ops
    name, N, H, W, Cout, k, stride, pad, pad_mode, dil, in_relu = case
    x = _rand(N, H, W, 4, seed=1).to(dev)
    x[..., 2:] = 0 if "flow" in name else x[..., 2:]              # the flow encoders' two padding channels are zeros
    w = _rand(Cout, 4, k, k, seed=2, scale=1.0 / math.sqrt(4 * k * k)).to(dev)
    b = _rand(Cout, seed=3).to(dev)
    pc = ops.PackedConv(w, b)
    xi = _hl(F.relu(x) if in_relu else x).permute(0, 3, 1, 2)
    if pad_mode == "replicate":
        xi, p_ = F.pad(xi, (pad, pad, pad, pad), mode="replicate"), 0
    else:
        p_ = pad
    ref = F.leaky_relu(F.conv2d(xi, _hl(w), b.double(), stride, p_, dil), 0.2).permute(0, 2, 3, 1)
    scale = max(ref.abs().max().item(), 1.0)
    kw = dict(stride=stride, pad=pad, dil=dil, pad_mode=pad_mode, in_relu=in_relu, act="lrelu", precision="bf16x3")
    other = ops.conv2d(x, pc, tile="128x128", **kw)               # an explicit tile: the register-staged kernel (conv_igemm.hip)
    got = ops.conv2d(x, pc, tile="c4", **kw)                      # csrc/conv_c4.hip
    e_other, e = (other.double() - ref).abs().max().item(), (got.double() - ref).abs().max().item()
    assert e <= max(2.0 * e_other, 2e-6 * scale) and e <= 2e-5 * scale, f"{name}: {e:.3e} vs the register-staged kernel's {e_other:.3e}"
    assert torch.equal(got, other), "the autotuner may pick conv_c4 next to the register-staged tiles only while they agree bit for bit"
    # outputs: fp32 + split (planes and interleaved) from one launch agree with the fp32 result
    o32, osp = ops.conv2d(x, pc, tile="c4", out_split="both", **kw)
    assert torch.equal(o32, got) and torch.equal(osp.data, ops.split(o32).data)
    if Cout % 32 == 0:
        oil = ops.conv2d(x, pc, tile="c4", out_split="only", out_il=True, **kw)
        assert oil.il and torch.equal(oil.data, ops.split(o32, interleave=True).data)
    print(f"[parity] conv_c4 {name}: max |conv_c4 - fp64| {e:.2e}, |conv_igemm - fp64| {e_other:.2e}, between them {(got - other).abs().max().item():.2e} (outputs up to {scale:.2f})")


def test_conv_c4_declines_other_layers(dev):
    from fgt_amd import ops
    x4, x8 = _rand(1, 16, 20, 4, seed=1).to(dev), _rand(1, 16, 20, 8, seed=1).to(dev)
    pc8 = ops.PackedConv(_rand(64, 8, 3, 3, seed=2, scale=0.2).to(dev), None)
    pc1 = ops.PackedConv(_rand(64, 4, 1, 1, seed=2, scale=0.2).to(dev), None)
    pc4 = ops.PackedConv(_rand(64, 4, 3, 3, seed=2, scale=0.2).to(dev), None)
    for xx, pp, kw in ((x8, pc8, dict(pad=1, precision="bf16x3")), (x4, pc1, dict(precision="bf16x3")), (x4, pc4, dict(pad=1, upsample=True, precision="bf16x3")),
                       (x4, pc4, dict(pad=1, precision="fp32"))):
        with pytest.raises(RuntimeError, match="does not serve"):
            ops.conv2d(xx, pp, tile="c4", **kw)
