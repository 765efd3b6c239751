"""Per-kernel parity on a real MI355X: every C-ABI entry point vs a plain fp32 PyTorch CPU restatement
of the reference op it replaces (same seeded inputs).  Tolerances are stated per test; fp32-MFMA kernels are
held to ~1e-5 relative to the output scale."""
import math

import pytest
import torch
import torch.nn.functional as F

from util import max_err, rel_err, report

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


CONV_CASES = [
    # name, N, H, W, Cin, Cout, k, stride, pad, dil, groups
    ("3x3_s1", 2, 20, 28, 64, 128, 3, 1, 1, 1, 1),
    ("3x3_s2_cin4", 2, 24, 40, 4, 64, 3, 2, 1, 1, 1),
    ("3x3_cout_odd", 1, 17, 23, 32, 126, 3, 1, 1, 1, 1),
    ("7x7_s3_p3", 2, 24, 36, 40, 96, 7, 3, 3, 1, 1),
    ("3x3_dil8", 1, 30, 27, 48, 48, 3, 1, 8, 8, 1),
    ("1x1_linear", 1, 1, 700, 512, 1960, 1, 1, 0, 1, 1),
    ("3x3_cout3", 1, 32, 48, 64, 3, 3, 1, 1, 1, 1),
    ("g4", 1, 15, 27, 64, 96, 3, 1, 1, 1, 4),
    ("1x5", 1, 20, 30, 64, 64, (1, 5), 1, (0, 2), 1, 1),
    ("5x1", 1, 20, 30, 64, 64, (5, 1), 1, (2, 0), 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize("tile", ["auto", "128x128", "128x64", "64x64", "128x32", "256x128"])
def test_conv2d_matches_torch(case, tile, dev):
    from fgt_amd import ops
    name, N, H, W, Cin, Cout, k, s, p, d, g = case
    kh, kw = (k, k) if isinstance(k, int) else k
    x = _rand(N, Cin, H, W, seed=1)
    w = _rand(Cout, Cin // g, kh, kw, seed=2, scale=1.0 / math.sqrt(Cin // g * kh * kw))
    b = _rand(Cout, seed=3)
    ref = F.leaky_relu(F.conv2d(x, w, b, s, p, d, g), 0.2)
    pc = ops.PackedConv(w.to(dev), b.to(dev), groups=g)
    out = ops.conv2d(nhwc(x).to(dev), pc, stride=s, pad=p, dil=d, act="lrelu", tile=tile)
    torch.cuda.synchronize()
    e, r = report(f"conv {name} tile={tile}", nchw(out.cpu()), ref)
    assert r < 2e-5


def test_conv2d_two_source_grouped_concat(dev):
    """Encoder layers 10-16: group-interleaved concat of x0 and out (FGT/models/model.py:57-66)."""
    from fgt_amd import ops
    N, H, W, g = 2, 15, 27, 8
    x0, o = _rand(N, 256, H, W, seed=1), _rand(N, 384, H, W, seed=2)
    w, b = _rand(256, 640 // g, 3, 3, seed=3, scale=0.05), _rand(256, seed=4)
    cat = torch.cat([x0.view(N, g, -1, H, W), o.view(N, g, -1, H, W)], 2).view(N, -1, H, W)
    ref = F.leaky_relu(F.conv2d(cat, w, b, 1, 1, 1, g), 0.2)
    pc = ops.PackedConv(w.to(dev), b.to(dev), groups=g)
    out = ops.conv2d(nhwc(x0).to(dev), pc, x1=nhwc(o).to(dev), stride=1, pad=1, act="lrelu")
    e, r = report("conv two-source g8", nchw(out.cpu()), ref)
    assert r < 2e-5


def test_conv2d_upsample_replicate_inrelu_and_slices(dev):
    from fgt_amd import ops
    N, H, W = 1, 12, 20
    x = _rand(N, 32, H, W, seed=1)
    w, b = _rand(48, 32, 3, 3, seed=2, scale=0.1), _rand(48, seed=3)
    pc = ops.PackedConv(w.to(dev), b.to(dev))
    # nearest x2 upsample then conv (network_blocks_2d.py:46-60)
    ref = F.leaky_relu(F.conv2d(F.interpolate(x, scale_factor=2), w, b, 1, 1), 0.2)
    out = ops.conv2d(nhwc(x).to(dev), pc, stride=1, pad=1, upsample=True, act="lrelu")
    assert report("conv upsample", nchw(out.cpu()), ref)[1] < 2e-5
    # replicate pad (model.py:207) 5x5
    w5, x2 = _rand(64, 2, 5, 5, seed=4, scale=0.2), _rand(2, 2, 16, 24, seed=5)
    ref = F.conv2d(F.pad(x2, (2, 2, 2, 2), mode="replicate"), w5, None)
    pc5 = ops.PackedConv(w5.to(dev), None)
    x2p = torch.zeros(2, 16, 24, 4)
    x2p[..., :2] = nhwc(x2)
    out = ops.conv2d(x2p.to(dev), pc5, stride=1, pad=2, pad_mode="replicate")
    assert report("conv replicate 5x5 cin2->4", nchw(out.cpu()), ref)[1] < 2e-5
    # ReLU on the gathered input + residual epilogue, reading/writing channel slices of wider buffers
    wide = _rand(N, H, W, 64, seed=6).to(dev)
    res = _rand(N, H, W, 48, seed=7)
    obuf = torch.zeros(N, H, W, 96, device=dev)
    ref = F.conv2d(F.relu(nchw(wide.cpu())[:, 16:48]), w, b, 1, 1) + nchw(res)
    ops.conv2d(wide[..., 16:48], pc, stride=1, pad=1, in_relu=True, epi="add", aux1=res.to(dev), out=obuf[..., 32:80])
    assert report("conv in_relu+add slices", nchw(obuf[..., 32:80].cpu()), ref)[1] < 2e-5
    assert obuf[..., :32].abs().max().item() == 0 and obuf[..., 80:].abs().max().item() == 0


def test_conv2d_epilogues(dev):
    from fgt_amd import ops
    rows = 300
    x, w, b = _rand(rows, 128, seed=1), _rand(64, 128, seed=2, scale=0.1), _rand(64, seed=3)
    a1, a2 = torch.sigmoid(_rand(rows, 64, seed=4)), _rand(rows, 64, seed=5)
    pc = ops.PackedConv(w.to(dev), b.to(dev))
    y = F.linear(x, w, b)
    out = ops.linear(x.to(dev), pc, act="sigmoid", epi="mul", aux1=a2.to(dev))
    assert report("epi mul", out.cpu(), torch.sigmoid(y) * a2)[1] < 2e-5
    out = ops.linear(x.to(dev), pc, act="relu", epi="add", aux1=a2.to(dev), act2="relu")
    assert report("epi add+relu", out.cpu(), F.relu(F.relu(y) + a2))[1] < 2e-5
    out = ops.linear(x.to(dev), pc, act="tanh", epi="gru", aux1=a1.to(dev), aux2=a2.to(dev))
    assert report("epi gru", out.cpu(), (1 - a1) * a2 + a1 * torch.tanh(y))[1] < 2e-5
    out = ops.linear(x.to(dev), pc, out_scale=0.25)
    assert report("out_scale", out.cpu(), 0.25 * y)[1] < 2e-5
    sc = _rand(64, seed=6).abs() + 0.5
    pcs = ops.PackedConv(w.to(dev), (b * sc).to(dev), scale=sc.to(dev))
    out = ops.linear(x.to(dev), pcs, act="relu")
    assert report("cscale (BN-eval)", out.cpu(), F.relu(y * sc))[1] < 2e-5
    # NCHW + tanh output (decoder.final)
    xi, wi = _rand(2, 16, 10, 14, seed=7), _rand(3, 16, 3, 3, seed=8, scale=0.2)
    pci = ops.PackedConv(wi.to(dev), None)
    out = ops.conv2d(nhwc(xi).to(dev), pci, stride=1, pad=1, act="tanh", out_nchw=True)
    assert report("nchw tanh", out.cpu(), torch.tanh(F.conv2d(xi, wi, None, 1, 1)))[1] < 2e-5


def test_layernorm(dev):
    from fgt_amd import ops
    rows = 333
    x0, x1 = _rand(rows, 512, seed=1), _rand(rows, 256, seed=2) * 3 + 1
    gA, bA, gB, bB = _rand(768, seed=3), _rand(768, seed=4), _rand(768, seed=5), _rand(768, seed=6)
    cat = torch.cat([x0, x1], 1)
    a, b = ops.layernorm(x0.to(dev), gA.to(dev), bA.to(dev), x1=x1.to(dev), gB=gB.to(dev), bB=bB.to(dev))
    assert report("ln two-source A", a.cpu(), F.layer_norm(cat, (768,), gA, bA))[0] < 2e-5
    assert report("ln two-source B", b.cpu(), F.layer_norm(cat, (768,), gB, bB))[0] < 2e-5
    o = ops.layernorm(x0.to(dev), gA[:512].to(dev), bA[:512].to(dev))
    assert report("ln 512", o.cpu(), F.layer_norm(x0, (512,), gA[:512], bA[:512]))[0] < 2e-5


def _sdpa(q, k, v):
    s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(q.size(-1))
    return torch.matmul(F.softmax(s, dim=-1), v)


@pytest.mark.parametrize("b,t,nh,nw", [(1, 3, 20, 36), (2, 2, 6, 8), (1, 5, 22, 36)])
def test_attention_temporal(b, t, nh, nw, dev):
    """attention_base.py:61-69: zone partition + SDPA + merge, q/k/v read in place from a fused buffer."""
    from fgt_amd import ops
    heads, G, c = 4, 2, 512
    qkv = _rand(b * t * nh * nw, 3 * c, seed=nh)
    zh, zw = nh // G, nw // G

    def zones(y):
        return y.view(b, t, G, zh, G, zw, heads, c // heads).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(b, G * G, heads, -1, c // heads)

    a = _sdpa(zones(qkv[:, :c] * 2), zones(qkv[:, c:2 * c] * 2), zones(qkv[:, 2 * c:]))
    ref = a.view(b, G, G, heads, t, zh, zw, c // heads).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(b * t * nh * nw, c)
    dq = qkv.clone()
    dq[:, :2 * c] *= 2          # larger logits -> peaky softmax, exercises the running-max rescale
    out = ops.attention_temporal(dq.to(dev), b, t, nh, nw, heads, G, c)
    assert report(f"attn temporal b{b} t{t} {nh}x{nw}", out.cpu(), ref)[1] < 2e-5


@pytest.mark.parametrize("bt,h,w", [(2, 20, 36), (1, 22, 35), (2, 8, 8)])
def test_attention_spatial(bt, h, w, dev):
    """attention_flow.py:98-110: windows + shared global tokens, crop on store."""
    from fgt_amd import ops
    heads, ws, gd, c = 4, 8, 4, 512
    nh, nw = (h + ws - 1) // ws * ws, (w + ws - 1) // ws * ws
    gh, gw = nh // ws, nw // ws
    ng = (nh // gd) * (nw // gd)
    q, k, v = (_rand(bt * nh * nw, c, seed=s) for s in (1, 2, 3))
    kg, vg = _rand(bt * ng, c, seed=4), _rand(bt * ng, c, seed=5)

    def windows(y):
        return y.view(bt, gh, ws, gw, ws, c).transpose(2, 3).reshape(bt, gh * gw, ws * ws, c)

    def split(y):
        return y.reshape(bt, gh * gw, -1, heads, c // heads).permute(0, 1, 3, 2, 4)

    K = torch.cat([windows(k), kg.view(bt, 1, ng, c).expand(-1, gh * gw, -1, -1)], 2)
    V = torch.cat([windows(v), vg.view(bt, 1, ng, c).expand(-1, gh * gw, -1, -1)], 2)
    a = _sdpa(split(windows(q)), split(K), split(V))
    a = a.transpose(2, 3).reshape(bt, gh, gw, ws, ws, c).transpose(2, 3).reshape(bt, nh, nw, c)[:, :h, :w].reshape(bt * h * w, c)
    out = ops.attention_spatial(q.to(dev), k.to(dev), v.to(dev), kg.to(dev), vg.to(dev), bt, h, w, nh, nw, heads, ws, ng)
    assert report(f"attn spatial bt{bt} {h}x{w}", out.cpu(), a)[1] < 2e-5


def test_attention_forced_rescale(dev):
    """A late key with a much larger logit forces the online-softmax rescale branch (flash softmax correctness)."""
    from fgt_amd import ops
    b, t, nh, nw, heads, G, c = 1, 4, 8, 8, 4, 2, 512
    qkv = _rand(b * t * nh * nw, 3 * c, seed=9) * 0.3
    qkv[-1, c:2 * c] = qkv[5, :c] * 40.0      # last token's key aligned with query 5
    zh, zw = nh // G, nw // G

    def zones(y):
        return y.view(b, t, G, zh, G, zw, heads, c // heads).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(b, G * G, heads, -1, c // heads)

    a = _sdpa(zones(qkv[:, :c]), zones(qkv[:, c:2 * c]), zones(qkv[:, 2 * c:]))
    ref = a.view(b, G, G, heads, t, zh, zw, c // heads).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(-1, c)
    out = ops.attention_temporal(qkv.to(dev), b, t, nh, nw, heads, G, c)
    assert report("attn forced rescale", out.cpu(), ref)[1] < 2e-5


def test_dw_pool_and_posemb(dev):
    from fgt_amd import ops
    bt, nh, nw, gd = 2, 24, 40, 4
    x0, x1 = _rand(bt, 512, nh, nw, seed=1), _rand(bt, 256, nh, nw, seed=2)
    w, b = _rand(768, 1, gd, gd, seed=3), _rand(768, seed=4)
    ref = F.conv2d(torch.cat([x0, x1], 1), w, b, stride=gd, groups=768).permute(0, 2, 3, 1).reshape(-1, 768)
    out = torch.empty(bt * (nh // gd) * (nw // gd), 768, device=dev)
    ops.dw_pool(nhwc(x0).view(-1, 512).to(dev), nhwc(x1).view(-1, 256).to(dev), bt, nh, nw, gd, w.to(dev), b.to(dev), out)
    assert report("dw_pool", out.cpu(), ref)[0] < 2e-5
    h, wd = 20, 36
    x = _rand(bt, 512, h, wd, seed=5)
    w3, b3 = _rand(512, 1, 3, 3, seed=6), _rand(512, seed=7)
    ref = nhwc(F.conv2d(x, w3, b3, 1, 1, 1, 512) + x)
    out = ops.dw3x3_residual(nhwc(x).to(dev), bt, h, wd, w3.to(dev), b3.to(dev))
    assert report("dw3x3+id", out.cpu(), ref)[0] < 2e-5


@pytest.mark.parametrize("Hf,Wf", [(60, 108), (64, 105), (12, 20)])
def test_fold(Hf, Wf, dev):
    """ffn_base.py:56-75 (normalised) and model.py:102-110 (plain) folds, tap-major columns."""
    from fgt_amd import ops
    k, s, p, Cc, fr = 7, 3, 3, 40, 2
    th, tw = (Hf + 2 * p - k) // s + 1, (Wf + 2 * p - k) // s + 1
    Y = _rand(fr, th * tw, Cc * k * k, seed=Hf)            # reference column order (c, ky, kx)
    folded = F.fold(Y.permute(0, 2, 1), (Hf, Wf), k, stride=s, padding=p)
    cnt = F.fold(torch.ones(fr, k * k, th * tw), (Hf, Wf), k, stride=s, padding=p)
    Yt = Y.view(fr, th * tw, Cc, k * k).permute(0, 1, 3, 2).reshape(fr * th * tw, k * k * Cc)   # tap-major
    out = ops.fold(Yt.to(dev), fr, th, tw, Cc, k, s, p, Hf, Wf, normalize=True)
    assert report("fold normalised", nchw(out.cpu()), folded / cnt)[1] < 1e-5
    res = _rand(fr, Hf, Wf, Cc, seed=3)
    out = ops.fold(Yt.to(dev), fr, th, tw, Cc, k, s, p, Hf, Wf, normalize=False, res=res.to(dev))
    assert report("fold + residual", nchw(out.cpu()), folded + nchw(res))[1] < 1e-5


def test_layout_and_pad(dev):
    from fgt_amd import ops
    x = _rand(3, 3, 10, 14, seed=1)
    m = _rand(3, 1, 10, 14, seed=2)
    buf = torch.full((3, 10, 14, 4), 7.0, device=dev)
    ops.nchw_to_nhwc(x.to(dev), buf, coff=0)
    ops.nchw_to_nhwc(m.to(dev), buf, coff=3)
    assert max_err(buf.cpu(), nhwc(torch.cat([x, m], 1))) == 0
    f = _rand(3, 2, 10, 14, seed=3)
    ops.nchw_to_nhwc(f.to(dev), buf, coff=0, zero_to=4)
    assert max_err(buf.cpu()[..., :2], nhwc(f)) == 0 and buf[..., 2:].abs().max().item() == 0
    assert max_err(ops.nhwc_to_nchw(buf[..., :2]).cpu(), f) == 0
    t = _rand(2 * 5 * 7, 8, seed=4)
    pd = ops.pad_tokens(t.to(dev), 2, 5, 7, 8, 8)
    ref = F.pad(t.view(2, 5, 7, 8), (0, 0, 0, 1, 0, 3)).reshape(-1, 8)
    assert max_err(pd.cpu(), ref) == 0
    back = ops.pad_tokens(pd, 2, 8, 8, 5, 7)
    assert max_err(back.cpu(), t) == 0
    a, b = _rand(50, 12, seed=5), _rand(50, 12, seed=6)
    assert max_err(ops.axpby(a.to(dev), 2.0, b.to(dev), -0.5, act="tanh").cpu(), torch.tanh(2 * a - 0.5 * b)) < 1e-6


DIRECT_CASES = [
    # name, Cin, Cout, k, act, epi, nchw
    ("fgt_final_64to3", 64, 3, 3, "tanh", None, True),
    ("lafc_flow_24to2", 24, 2, 3, None, None, False),
    ("edge_16to1_1x1", 16, 1, 1, "sigmoid", None, True),
    ("raft_flowhead_256to2_add", 256, 2, 3, None, "add", False),
    ("cin4_to4", 4, 4, 3, "lrelu", None, False),
]


@pytest.mark.parametrize("case", DIRECT_CASES, ids=[c[0] for c in DIRECT_CASES])
def test_conv2d_direct_small_cout(case, dev):
    """Cout <= 4 layers take the VALU direct-conv kernel when tile='auto'; the MFMA tile path must agree."""
    from fgt_amd import ops
    name, Cin, Cout, k, act, epi, nchw_out = case
    N, H, W = 2, 19, 27
    x = _rand(N, Cin, H, W, seed=1)
    w, b = _rand(Cout, Cin, k, k, seed=2, scale=1.0 / math.sqrt(Cin * k * k)), _rand(Cout, seed=3)
    y = F.conv2d(x, w, b, 1, k // 2)
    y = {"tanh": torch.tanh, "sigmoid": torch.sigmoid, "lrelu": lambda v: F.leaky_relu(v, 0.2), None: lambda v: v}[act](y)
    aux = _rand(N, H, W, Cout, seed=4) if epi else None
    if epi:
        y = y + nchw(aux)
    pc = ops.PackedConv(w.to(dev), b.to(dev))
    for tile in ("auto", "128x32"):
        out = ops.conv2d(nhwc(x).to(dev), pc, stride=1, pad=k // 2, act=act, epi=epi, aux1=None if aux is None else aux.to(dev),
                         out_nchw=nchw_out, tile=tile)
        got = out.cpu() if nchw_out else nchw(out.cpu())
        assert report(f"direct conv {name} tile={tile}", got, y)[1] < 2e-5


BF16X3_CASES = [c for c in CONV_CASES if c[0] in ("3x3_s1", "3x3_s2_cin4", "7x7_s3_p3", "1x1_linear", "g4", "3x3_cout_odd")]


@pytest.mark.parametrize("case", BF16X3_CASES, ids=[c[0] for c in BF16X3_CASES])
@pytest.mark.parametrize("tile", ["128x128", "128x64", "64x64", "128x32", "256x128"])
def test_conv2d_bf16x3_split_precision(case, tile, dev):
    """hi/lo bf16 split (3 bf16 MFMAs per product, fp32 accumulate): ~2^-16 relative error per product."""
    from fgt_amd import ops
    name, N, H, W, Cin, Cout, k, s, p, d, g = case
    kh, kw = (k, k) if isinstance(k, int) else k
    x = _rand(N, Cin, H, W, seed=1)
    w = _rand(Cout, Cin // g, kh, kw, seed=2, scale=1.0 / math.sqrt(Cin // g * kh * kw))
    b = _rand(Cout, seed=3)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), s, p, d, g), 0.2).float()
    pc = ops.PackedConv(w.to(dev), b.to(dev), groups=g)
    out = ops.conv2d(nhwc(x).to(dev), pc, stride=s, pad=p, dil=d, act="lrelu", tile=tile, precision="bf16x3")
    e, r = report(f"conv bf16x3 {name} tile={tile}", nchw(out.cpu()), ref)
    assert r < 5e-5


@pytest.mark.parametrize("b,t,nh,nw", [(1, 3, 20, 36), (2, 2, 6, 8), (1, 5, 22, 36)])
def test_attention_temporal_bf16x3(b, t, nh, nw, dev):
    """Same contract as test_attention_temporal with Q/K/P/V split into hi/lo bf16 (3 bf16 MFMAs per product)."""
    from fgt_amd import ops
    heads, G, c = 4, 2, 512
    qkv = _rand(b * t * nh * nw, 3 * c, seed=nh)
    qkv[:, :2 * c] *= 2
    zh, zw = nh // G, nw // G

    def zones(y):
        return y.view(b, t, G, zh, G, zw, heads, c // heads).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(b, G * G, heads, -1, c // heads)

    a = _sdpa(zones(qkv[:, :c].double()), zones(qkv[:, c:2 * c].double()), zones(qkv[:, 2 * c:].double())).float()
    ref = a.view(b, G, G, heads, t, zh, zw, c // heads).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(b * t * nh * nw, c)
    out = ops.attention_temporal(qkv.to(dev), b, t, nh, nw, heads, G, c, precision="bf16x3")
    assert report(f"attn temporal bf16x3 b{b} t{t} {nh}x{nw}", out.cpu(), ref)[1] < 1e-4


@pytest.mark.parametrize("bt,h,w", [(2, 20, 36), (1, 22, 35)])
def test_attention_spatial_bf16x3(bt, h, w, dev):
    from fgt_amd import ops
    heads, ws, gd, c = 4, 8, 4, 512
    nh, nw = (h + ws - 1) // ws * ws, (w + ws - 1) // ws * ws
    gh, gw = nh // ws, nw // ws
    ng = (nh // gd) * (nw // gd)
    q, k, v = (_rand(bt * nh * nw, c, seed=s) for s in (1, 2, 3))
    kg, vg = _rand(bt * ng, c, seed=4), _rand(bt * ng, c, seed=5)

    def windows(y):
        return y.view(bt, gh, ws, gw, ws, c).transpose(2, 3).reshape(bt, gh * gw, ws * ws, c)

    def split(y):
        return y.reshape(bt, gh * gw, -1, heads, c // heads).permute(0, 1, 3, 2, 4)

    K = torch.cat([windows(k), kg.view(bt, 1, ng, c).expand(-1, gh * gw, -1, -1)], 2)
    V = torch.cat([windows(v), vg.view(bt, 1, ng, c).expand(-1, gh * gw, -1, -1)], 2)
    a = _sdpa(split(windows(q)).double(), split(K).double(), split(V).double()).float()
    a = a.transpose(2, 3).reshape(bt, gh, gw, ws, ws, c).transpose(2, 3).reshape(bt, nh, nw, c)[:, :h, :w].reshape(bt * h * w, c)
    out = ops.attention_spatial(q.to(dev), k.to(dev), v.to(dev), kg.to(dev), vg.to(dev), bt, h, w, nh, nw, heads, ws, ng,
                                precision="bf16x3")
    assert report(f"attn spatial bf16x3 bt{bt} {h}x{w}", out.cpu(), a)[1] < 1e-4


def test_attention_bf16x3_forced_rescale(dev):
    from fgt_amd import ops
    b, t, nh, nw, heads, G, c = 1, 4, 8, 8, 4, 2, 512
    qkv = _rand(b * t * nh * nw, 3 * c, seed=9) * 0.3
    qkv[-1, c:2 * c] = qkv[5, :c] * 40.0
    zh, zw = nh // G, nw // G

    def zones(y):
        return y.view(b, t, G, zh, G, zw, heads, c // heads).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(b, G * G, heads, -1, c // heads)

    a = _sdpa(zones(qkv[:, :c].double()), zones(qkv[:, c:2 * c].double()), zones(qkv[:, 2 * c:].double())).float()
    ref = a.view(b, G, G, heads, t, zh, zw, c // heads).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(-1, c)
    out = ops.attention_temporal(qkv.to(dev), b, t, nh, nw, heads, G, c, precision="bf16x3")
    assert report("attn bf16x3 forced rescale", out.cpu(), ref)[1] < 1e-4


@pytest.mark.parametrize("Cin,Cout,act,epi,nchw_out,two_src", [(64, 3, "tanh", None, True, False), (256, 2, None, "add", False, False),
                                                                (48, 1, "sigmoid", None, False, True), (16, 4, "lrelu", None, False, False)])
def test_conv3x3_lds_tiled_small_cout(Cin, Cout, act, epi, nchw_out, two_src, dev):
    """3x3/s1/p1, Cout <= 4, Cin % 16 == 0 and a map of at least 8 x 32: the LDS-tiled direct kernel (partial edge tiles)."""
    from fgt_amd import ops
    N, H, W = 2, 21, 45
    x = _rand(N, Cin, H, W, seed=1)
    w, b = _rand(Cout, Cin, 3, 3, seed=2, scale=1.0 / math.sqrt(Cin * 9)), _rand(Cout, seed=3)
    y = F.conv2d(x, w, b, 1, 1)
    y = {"tanh": torch.tanh, "sigmoid": torch.sigmoid, "lrelu": lambda v: F.leaky_relu(v, 0.2), None: lambda v: v}[act](y)
    aux = _rand(N, H, W, Cout, seed=4) if epi else None
    if epi:
        y = y + nchw(aux)
    pc = ops.PackedConv(w.to(dev), b.to(dev))
    xh = nhwc(x).to(dev)
    kw = dict(x1=xh[..., 32:].contiguous()) if two_src else {}
    x0 = xh[..., :32].contiguous() if two_src else xh
    out = ops.conv2d(x0, pc, stride=1, pad=1, act=act, epi=epi, aux1=None if aux is None else aux.to(dev), out_nchw=nchw_out, **kw)
    got = out.cpu() if nchw_out else nchw(out.cpu())
    assert report(f"tiled 3x3 {Cin}->{Cout}", got, y)[1] < 2e-5


# ---------------------------------------------------------------------------------------------------------------------------
# Long temporal zones: the launcher switches to attn_bf16x3_kernel<8, true> (8 wavefronts per K/V tile, register prefetch) at
# n_q >= 2048 — every window of the 432x240x80 bench clip (t = 13 / 17 / 18 on the 20x36 grid: L = 2340 / 3060 / 3240) and the
# 864x480x160 config (t = 26 on 40x72: L = 18720).  Reference: fp64 softmax(QK^T/sqrt(d))V per (zone, head) — on the CPU for
# t = 13, with torch fp64 on the GPU for the larger ones (an [L, L] fp64 score matrix per problem; independent of libfgt_hip).
def _temporal_ref64(qkv, b, t, nh, nw, heads, G, c, device):
    zh, zw, dh = nh // G, nw // G, c // heads
    x = qkv.to(device).double().view(b, t, G, zh, G, zw, 3, heads, dh)
    out = torch.empty(b, t, G, zh, G, zw, heads, dh, dtype=torch.float32)
    for bi in range(b):
        for zi in range(G):
            for zj in range(G):
                for hd in range(heads):
                    q, k, v = (x[bi, :, zi, :, zj, :, j, hd].reshape(-1, dh) for j in range(3))
                    s = (q @ k.T) / math.sqrt(dh)
                    o = torch.softmax(s, dim=-1) @ v
                    out[bi, :, zi, :, zj, :, hd] = o.view(t, zh, zw, dh).float().cpu()
                    del s
    return out.reshape(b * t * nh * nw, c)


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("bf16x3", 1e-4)])
@pytest.mark.parametrize("b,t,nh,nw", [(1, 13, 20, 36), (1, 18, 20, 36), (2, 17, 20, 36), (1, 26, 40, 72)])
def test_attention_temporal_long_zones(b, t, nh, nw, prec, tol, dev):
    """attention_base.py:16-22 at the zone lengths the benchmark runs: both precisions, incl. split output."""
    from fgt_amd import ops
    heads, G, c = 4, 2, 512
    L = t * (nh // G) * (nw // G)
    assert L >= 2048                                  # the 8-wavefront bf16x3 instance / the 4-wavefront fp32 one with many key tiles
    qkv = _rand(b * t * nh * nw, 3 * c, seed=100 + t)
    qkv[:, :2 * c] *= 1.5                             # logits ~ N(0, 2.25 * sqrt(128)/sqrt(128)): peaky rows, running max moves often
    ref = _temporal_ref64(qkv, b, t, nh, nw, heads, G, c, "cpu" if t == 13 else dev)
    dq = qkv.to(dev)
    out = ops.attention_temporal(dq, b, t, nh, nw, heads, G, c, precision=prec)
    assert report(f"attn temporal long {prec} b{b} t{t} {nh}x{nw} (L={L})", out.cpu(), ref)[1] < tol
    if prec == "bf16x3":                              # the path the model uses: output handed over pre-split
        sp = ops.attention_temporal(dq, b, t, nh, nw, heads, G, c, precision=prec, out_split=True)
        hi, lo = sp.planes()
        assert torch.equal(hi.float() + lo.float(), (lambda o: o.to(torch.bfloat16).float() + (o - o.to(torch.bfloat16).float()).to(torch.bfloat16).float())(out))


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("bf16x3", 1e-4)])
def test_attention_spatial_config5_grid(prec, tol, dev):
    """attention_flow.py:98-110 on the 864x480 token grid (40x72 -> 45 windows of 64 queries x (64 + 180) keys per frame)."""
    from fgt_amd import ops
    bt, h, w, heads, ws, gd, c = 3, 40, 72, 4, 8, 4, 512
    nh, nw = h, w
    gh, gw, ng = nh // ws, nw // ws, (nh // gd) * (nw // gd)
    q, k, v = (_rand(bt * nh * nw, c, seed=s) for s in (11, 12, 13))
    kg, vg = _rand(bt * ng, c, seed=14), _rand(bt * ng, c, seed=15)
    windows = lambda y: y.view(bt, gh, ws, gw, ws, c).transpose(2, 3).reshape(bt, gh * gw, ws * ws, c)
    split = lambda y: y.reshape(bt, gh * gw, -1, heads, c // heads).permute(0, 1, 3, 2, 4)
    K = torch.cat([windows(k), kg.view(bt, 1, ng, c).expand(-1, gh * gw, -1, -1)], 2)
    V = torch.cat([windows(v), vg.view(bt, 1, ng, c).expand(-1, gh * gw, -1, -1)], 2)
    a = _sdpa(split(windows(q)).double(), split(K).double(), split(V).double()).float()
    a = a.transpose(2, 3).reshape(bt, gh, gw, ws, ws, c).transpose(2, 3).reshape(bt * h * w, c)
    out = ops.attention_spatial(q.to(dev), k.to(dev), v.to(dev), kg.to(dev), vg.to(dev), bt, h, w, nh, nw, heads, ws, ng, precision=prec)
    assert report(f"attn spatial 40x72 {prec}", out.cpu(), a)[1] < tol


# ---------------------------------------------------------------------------------------------------------------------------
# Split-input attention (csrc/attention_split.hip): q / k / v arrive as hi/lo bf16 planes from the projection GEMMs, K / V tiles are
# streamed by LDS-DMA and V is read through ds_read_b64_tr_b16.  Reference: fp64 attention on hi + lo (what the planes represent:
# the fp32 values rounded to 16 mantissa bits), so the tolerance is the same bf16x3 bar as above.
def _split_vals(sp):
    return sp.float().cpu()          # hi + lo as fp32


@pytest.mark.parametrize("b,t,nh,nw", [(1, 3, 20, 36), (2, 2, 6, 8), (1, 5, 22, 36), (1, 13, 20, 36), (2, 17, 20, 36), (1, 26, 40, 72)])
def test_attention_temporal_split_inputs(b, t, nh, nw, dev):
    from fgt_amd import ops
    heads, G, c = 4, 2, 512
    qkv = _rand(b * t * nh * nw, 3 * c, seed=200 + t)
    qkv[:, :2 * c] *= 1.5
    sp = ops.split(qkv.to(dev))
    ref = _temporal_ref64(_split_vals(sp), b, t, nh, nw, heads, G, c, "cpu" if t <= 5 else dev)
    out = ops.attention_temporal(sp, b, t, nh, nw, heads, G, c)
    assert report(f"attn temporal split-in b{b} t{t} {nh}x{nw}", out.cpu(), ref)[1] < 1e-4
    o2 = ops.attention_temporal(sp, b, t, nh, nw, heads, G, c, out_split=True)          # the model's path: split in, split out
    assert torch.equal(o2.float(), ops.split(out).float())
    # a Split that is a slice of a wider fused buffer with extra leading rows (ld > 3c, row offset)
    wide = ops.split(torch.cat([_rand(7, 3 * c + 64, seed=1), torch.cat([qkv, _rand(qkv.shape[0], 64, seed=2)], 1)], 0).to(dev))
    view = ops.Split(wide.data[:, 7:, :3 * c])
    assert torch.equal(ops.attention_temporal(view, b, t, nh, nw, heads, G, c), out)


@pytest.mark.parametrize("b,t,tq,nh,nw", [(1, 5, 3, 20, 36), (3, 4, 1, 6, 8), (2, 17, 11, 20, 36), (1, 26, 11, 40, 72)])
@pytest.mark.parametrize("kind", ["fp32", "bf16x3", "split"])
def test_attention_temporal_query_prefix(b, t, tq, nh, nw, kind, dev):
    """fgt_attn_desc.tq: queries of the first tq frames of every batch element only (keys / values: all t frames), O compact —
    the rows must be the very bits of the full call (HIP vs HIP; the full call is checked against fp64 above)."""
    from fgt_amd import ops
    heads, G, c = 4, 2, 512
    qkv = _rand(b * t * nh * nw, 3 * c, seed=300 + t).to(dev)
    src = ops.split(qkv) if kind == "split" else qkv
    prec = None if kind == "split" else kind
    full = ops.attention_temporal(src, b, t, nh, nw, heads, G, c, precision=prec)
    part = ops.attention_temporal(src, b, t, nh, nw, heads, G, c, precision=prec, tq=tq)
    n = nh * nw
    assert part.shape == (b * tq * n, c)
    assert torch.equal(part.view(b, tq * n, c), full.view(b, t * n, c)[:, : tq * n])
    if kind == "split":
        ps = ops.attention_temporal(src, b, t, nh, nw, heads, G, c, tq=tq, out_split=True)
        assert torch.equal(ps.float(), ops.split(part).float())


def test_attention_split_forced_rescale(dev):
    from fgt_amd import ops
    b, t, nh, nw, heads, G, c = 1, 4, 8, 8, 4, 2, 512
    qkv = _rand(b * t * nh * nw, 3 * c, seed=9) * 0.3
    qkv[-1, c:2 * c] = qkv[5, :c] * 40.0
    sp = ops.split(qkv.to(dev))
    ref = _temporal_ref64(_split_vals(sp), b, t, nh, nw, heads, G, c, "cpu")
    out = ops.attention_temporal(sp, b, t, nh, nw, heads, G, c)
    assert report("attn split-in forced rescale", out.cpu(), ref)[1] < 1e-4


@pytest.mark.parametrize("bt,h,w", [(2, 20, 36), (1, 22, 35), (2, 8, 8), (3, 40, 72)])
def test_attention_spatial_split_inputs(bt, h, w, dev):
    from fgt_amd import ops
    heads, ws, gd, c = 4, 8, 4, 512
    nh, nw = (h + ws - 1) // ws * ws, (w + ws - 1) // ws * ws
    gh, gw = nh // ws, nw // ws
    ng = (nh // gd) * (nw // gd)
    rows = bt * nh * nw
    # k / v as ONE buffer [window rows | global rows] like the model's fused projections
    qs = ops.split(_rand(rows, c, seed=31).to(dev))
    ks = ops.split(_rand(rows + bt * ng, c, seed=32).to(dev))
    vs = ops.split(_rand(rows + bt * ng, c, seed=33).to(dev))
    q, kall, vall = _split_vals(qs), _split_vals(ks), _split_vals(vs)
    k, kg, v, vg = kall[:rows], kall[rows:], vall[:rows], vall[rows:]
    windows = lambda y: y.view(bt, gh, ws, gw, ws, c).transpose(2, 3).reshape(bt, gh * gw, ws * ws, c)
    heads_ = lambda y: y.reshape(bt, gh * gw, -1, heads, c // heads).permute(0, 1, 3, 2, 4)
    K = torch.cat([windows(k), kg.view(bt, 1, ng, c).expand(-1, gh * gw, -1, -1)], 2)
    V = torch.cat([windows(v), vg.view(bt, 1, ng, c).expand(-1, gh * gw, -1, -1)], 2)
    a = _sdpa(heads_(windows(q)).double(), heads_(K).double(), heads_(V).double()).float()
    a = a.transpose(2, 3).reshape(bt, gh, gw, ws, ws, c).transpose(2, 3).reshape(bt, nh, nw, c)[:, :h, :w].reshape(bt * h * w, c)
    out = ops.attention_spatial(qs, ks[:rows], vs[:rows], ks[rows:], vs[rows:], bt, h, w, nh, nw, heads, ws, ng)
    assert report(f"attn spatial split-in bt{bt} {h}x{w}", out.cpu(), a)[1] < 1e-4


def test_mfma_probe_reports_a_plausible_sustained_rate(dev):
    """fgt_mfma_probe (bench.py's `roofline.sustained`): rates below the nominal peaks, clock within the part's range."""
    from fgt_amd import ops
    tf, ghz = ops.mfma_probe(f32=False, iters=4000, device=dev)
    print(f"[probe] bf16 MFMA sustained {tf:.0f} TFLOP/s at {ghz:.2f} GHz")
    assert 800 < tf < 2600 and 1.0 < ghz < 2.6
    tf32, ghz32 = ops.mfma_probe(f32=True, iters=1000, device=dev)
    print(f"[probe] fp32 MFMA sustained {tf32:.0f} TFLOP/s at {ghz32:.2f} GHz")
    assert 60 < tf32 < 165 and 1.0 < ghz32 < 2.6


# ---------------------------------------------------------------------------------------------------------------------------
# Compact token maps (fgt_attn_desc.compact / fgt_dw_pool's real-grid extents): the reference's zero-padded tokens are never
# materialised — every padded position reads ONE row.  HIP vs HIP: the compact call must give the bits of the call on padded maps
# (the padded calls are checked against fp64 / torch above).
def _compact(m, bt, h, w, nh, nw, pad):
    """[bt*nh*nw, c] padded map whose padded rows all equal `pad` -> ([bt*h*w + 1, c] compact map with the pad row last)."""
    c = m.shape[1]
    return torch.cat([m.view(bt, nh, nw, c)[:, :h, :w].reshape(bt * h * w, c), pad.view(1, c)], 0).contiguous()


@pytest.mark.parametrize("bt,h,w", [(2, 20, 36), (1, 22, 35), (3, 8, 8), (1, 40, 72)])
@pytest.mark.parametrize("kind", ["fp32", "bf16x3", "split"])
def test_attention_spatial_compact_maps(bt, h, w, kind, dev):
    from fgt_amd import ops
    heads, ws, gd, c = 4, 8, 4, 512
    nh, nw = (h + ws - 1) // ws * ws, (w + ws - 1) // ws * ws
    ng = (nh // gd) * (nw // gd)
    rows, R = bt * nh * nw, bt * h * w
    maps = []
    for i in range(3):                                    # q, k, v on the padded grid, every padded row = one constant row
        m = _rand(rows, c, seed=40 + i).view(bt, nh, nw, c)
        pad = _rand(c, seed=50 + i)
        m[:, h:] = pad
        m[:, :, w:] = pad
        maps.append((m.reshape(rows, c).to(dev), pad.to(dev)))
    kg, vg = _rand(bt * ng, c, seed=60).to(dev), _rand(bt * ng, c, seed=61).to(dev)
    prec = None if kind == "split" else kind
    wrap = (lambda t: ops.split(t)) if kind == "split" else (lambda t: t)
    padded = ops.attention_spatial(*(wrap(m) for m, _ in maps), wrap(kg), wrap(vg), bt, h, w, nh, nw, heads, ws, ng, precision=prec)
    comp = [wrap(_compact(m, bt, h, w, nh, nw, p)) for m, p in maps]
    got = ops.attention_spatial(*comp, wrap(kg), wrap(vg), bt, h, w, nh, nw, heads, ws, ng, precision=prec, pad_row=R)
    assert torch.equal(got, padded)


def test_dw_pool_compact_maps(dev):
    from fgt_amd import ops
    bt, h, w, nh, nw, gd = 2, 20, 36, 24, 40, 4
    x0, x1 = _rand(bt, h, w, 512, seed=1), _rand(bt, h, w, 256, seed=2)
    wt, b = _rand(768, 1, gd, gd, seed=3).to(dev), _rand(768, seed=4).to(dev)
    pad = lambda t: F.pad(t, (0, 0, 0, nw - w, 0, nh - h)).reshape(bt * nh * nw, -1).to(dev)
    want = torch.empty(bt * (nh // gd) * (nw // gd), 768, device=dev)
    ops.dw_pool(pad(x0), pad(x1), bt, nh, nw, gd, wt, b, want)
    got = torch.empty_like(want)
    ops.dw_pool(x0.reshape(bt * h * w, -1).to(dev), x1.reshape(bt * h * w, -1).to(dev), bt, nh, nw, gd, wt, b, got, h=h, w_real=w)
    assert torch.equal(got, want)
    got3 = torch.empty(bt * (nh // 2) * (nw // 2), 768, device=dev)            # generic (k != 4) kernel
    want3 = torch.empty_like(got3)
    w2 = _rand(768, 1, 2, 2, seed=5).to(dev)
    ops.dw_pool(pad(x0), pad(x1), bt, nh, nw, 2, w2, b, want3)
    ops.dw_pool(x0.reshape(bt * h * w, -1).to(dev), x1.reshape(bt * h * w, -1).to(dev), bt, nh, nw, 2, w2, b, got3, h=h, w_real=w)
    assert torch.equal(got3, want3)
