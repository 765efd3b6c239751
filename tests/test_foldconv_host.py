"""Host side of the fold convolution (fgt_amd.fgt_model.fold_conv_weight / fold_conv_tables / fold_conv_layout) without a GPU: the re-laid weights,
the per-position tables and the sub-pixel scatter, run over the executable kernel specification (tests/fake_ops.py), against the REFERENCE
formulation — nn.Linear, F.fold, division by F.fold(ones) (FGT/models/transformer_base/ffn_base.py:53-66) and Vec2Patch + residual
(FGT/models/model.py:102-110, 280) — on token grids that tile the feature map exactly and that do not (the tool's 256-high default: 22 x 3 = 66 > 64)."""
import math

import pytest
import torch
import torch.nn.functional as F

import fake_ops
from fgt_amd import fgt_model as M

torch.set_grad_enabled(False)

GEOMS = [
    # th, tw, Hf, Wf (Hf = H / 4 of the input; th = (Hf + 2*3 - 7) // 3 + 1)
    (20, 36, 60, 108),      # 240 x 432: exact
    (22, 36, 64, 108),      # 256 x 432 (the tool's default height): the last token row's sub-pixels 1, 2 fall off the map
    (6, 8, 16, 24), (4, 7, 12, 20), (1, 1, 3, 3), (2, 1, 4, 1),
]


@pytest.mark.parametrize("th,tw,Hf,Wf", GEOMS)
@pytest.mark.parametrize("cc,normalize", [(40, True), (128, False), (8, True)])
def test_fold_conv_host_side_equals_linear_fold(th, tw, Hf, Wf, cc, normalize):
    k, s, p, cin, N = 7, 3, 3, 32, 2
    assert (Hf + 2 * p - k) // s + 1 == th and (Wf + 2 * p - k) // s + 1 == tw
    g = torch.Generator().manual_seed(th * 100 + tw + cc)
    x = torch.randn(N * th * tw, cin, generator=g)
    w = torch.randn(cc * k * k, cin, generator=g) / math.sqrt(cin)
    b = torch.randn(cc * k * k, generator=g)
    res = None if normalize else torch.randn(N, Hf, Wf, cc, generator=g)
    # the reference formulation (fp64)
    y = (x.double() @ w.double().t() + b.double()).view(N, th * tw, -1).permute(0, 2, 1)
    ref = F.fold(y, (Hf, Wf), k, stride=s, padding=p)
    if normalize:
        ref = F.relu(ref / F.fold(torch.ones(N, k * k, th * tw, dtype=torch.float64), (Hf, Wf), k, stride=s, padding=p))
    ref = ref.permute(0, 2, 3, 1)
    if res is not None:
        ref = ref + res.double()
    # the fold convolution over the kernel specification
    assert M.fold_conv_supported(k, s, p) and not M.fold_conv_supported(7, 3, 2) and not M.fold_conv_supported(5, 3, 3)
    g0, cout, col0 = M.fold_conv_layout(cc, s)
    assert g0 % 128 == 0 and g0 >= s * cc and cout == g0 + (s - 1) * s * cc and col0(0, 0) == 0 and col0(1, 0) == g0
    W = M.fold_conv_weight(w, cc, k, s)
    assert W.shape == (cout, cin, 3, 3)
    assert float(W[g0:, :, 0].abs().max()) == 0.0, "the r_y >= 1 columns have no ky = 0 tap (ky_skip_n0)"
    assert float(W[s * cc:g0].abs().max() if g0 > s * cc else 0.0) == 0.0, "padding columns"
    assert int((W != 0).sum()) == w.numel(), "every weight of the Linear appears exactly once"
    off, sc = M.fold_conv_tables(b, cc, k, s, th, tw, normalize)
    pc = fake_ops.PackedConv(W, None)
    kw = dict(stride=1, pad=1, aux_per_image=True, ps=(s, cc, g0, Hf, Wf), ky_skip_n0=g0)
    kw.update(dict(act="relu", epi="affine", aux1=off, aux2=sc) if normalize else dict(epi="ps_add2", aux1=off, aux2=res))
    out = fake_ops.conv2d(x.view(N, th, tw, cin), pc, **kw)
    assert tuple(out.shape) == (N, Hf, Wf, cc)
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err
