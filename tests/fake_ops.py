"""Executable CPU specification of the kernel contracts in include/fgt_hip.h (TEST INFRASTRUCTURE ONLY).

Each function mirrors the signature of fgt_amd.ops and restates, with torch CPU ops, what the HIP kernel is
specified to compute (packed-weight layout, two-source concat, tap-major fold columns, zone/window addressing ...).
tests/test_host_logic.py monkeypatches it under the nn.Module mirrors to check the HOST logic (shapes, views,
weight re-layouts, call order) against the oracle without a GPU.  The product never imports this file.
"""
import math

import torch
import torch.nn.functional as F

from fgt_amd.ops import PackedConv, ceil_to, _as_map  # noqa: F401  (host-side helpers are device agnostic)

_ACT = {None: lambda v, s: v, "none": lambda v, s: v, "lrelu": lambda v, s: F.leaky_relu(v, s), "relu": lambda v, s: F.relu(v),
        "sigmoid": lambda v, s: torch.sigmoid(v), "tanh": lambda v, s: torch.tanh(v)}


DEFAULT_CONV_PRECISION = "fp32"      # tests flip it to "bf16x3" / "f16" to walk the split-chain plumbing of the host code


def _f16_mode():
    return DEFAULT_CONV_PRECISION == "f16"


def split_il(channels):
    return False                                  # layouts are not modelled on the CPU (ops.split_il)


def f16_round(x):
    """The fp16 activation / weight format of the 'f16' mode (csrc/common.h fgt_half4): f16_rne(clamp(x, +-65504)), as fp32 values."""
    return x.clamp(-65504.0, 65504.0).to(torch.float16).float()


class Split:
    """Stand-in for ops.Split: carries the fp32 tensor the split planes stand for (the CPU spec is exact; the 2^-16 rounding of the
    real format is a kernel property, pinned on the GPU in tests/test_split_gpu.py).  In the 'f16' mode the tensor holds the values
    ROUNDED to fp16 (h = True; writers go through `put`): there the rounding is 2^-11 and part of the specification — this file is
    then the numerical model of that mode (operands rounded once by their producer, exact products, fp32-or-better accumulation)."""

    def __init__(self, x, h=None):
        self.h = _f16_mode() if h is None else h
        self.x = f16_round(x) if self.h else x

    @staticmethod
    def empty(shape, device=None, interleaved=False, h=None):
        return Split(torch.zeros(tuple(shape)), h)           # (the hi/lo layout — planes or interleaved — is a kernel property: not modelled)

    def put(self, val):
        self.x.copy_(f16_round(val) if self.h else val)

    @property
    def shape(self):
        return self.x.shape

    def __getitem__(self, idx):
        s = Split.__new__(Split)            # a view of the same storage (writers fill slices in place), already rounded
        s.h, s.x = self.h, self.x[idx]
        return s

    def view(self, *shape):
        s = Split.__new__(Split)
        s.h, s.x = self.h, self.x.view(*shape)
        return s

    def channels(self, c0, c1):
        s = Split.__new__(Split)
        s.h, s.x = self.h, self.x[..., c0:c1]
        return s


def split(x, relu=False, out=None, interleave=False, h=None):
    assert not isinstance(x, Split), "double split"
    v = F.relu(x) if relu else x
    if out is not None:
        out.put(v.reshape(out.x.shape))
        return out
    return Split(v, h)


def _dense_weight(pc):
    G, Cout_g = pc.groups, pc.Cout // pc.groups
    w = pc.w[:, :Cout_g, :pc.K].reshape(G, Cout_g, pc.kh, pc.kw, pc.Cg).permute(0, 1, 4, 2, 3)
    return w.reshape(pc.Cout, pc.Cg, pc.kh, pc.kw)


def conv2d(x, pc, x1=None, stride=1, pad=0, dil=1, upsample=False, pad_mode="zeros", in_relu=False, act=None, slope=0.2,
           epi=None, aux1=None, aux2=None, act2=None, out_scale=1.0, out=None, out_nchw=False, tile=None, precision=None,
           out_split=None, out_s=None, out_il=False, out_h=None, ps=None, ky_skip_n0=0, aux_per_image=False, n_alg=0, bias_map=None, tile_order=0,
           dual=False):
    """ps = (r, c, g0, Hf, Wf): the sub-pixel output of fgt_conv_desc.ps_r (fold as a convolution); ky_skip_n0 / n_alg change no value;
    bias_map replaces pc.bias by an [N, Ho, Wo, Cout] map (fgt_conv_desc.ld_bias).
    dual (fgt_conv_desc.dual_n0 = Cout / 2): head 0 = act(v) of the first half of the columns (fp32), head 1 = act(v) * aux1 of the second (Split)."""
    if dual:
        assert epi == "mul" and out_split == "both" and ps is None and pc.groups == 1
        y = conv2d(x, pc, x1, stride, pad, dil, upsample, pad_mode, in_relu, act, slope, None, None, None, act2, out_scale, None, bias_map=bias_map)
        h = pc.Cout // 2
        y0 = y[..., :h].contiguous()
        y1 = y[..., h:] * aux1.reshape(y.shape[0], y.shape[1], y.shape[2], h)
        if out is not None:
            out.copy_(y0.reshape(out.shape))
            y0 = out
        if out_s is not None:
            out_s.put(y1.reshape(out_s.x.shape))
            return y0, out_s
        return y0, Split(y1.contiguous(), out_h)
    if out_split:
        assert not out_nchw
        y = conv2d(x, pc, x1, stride, pad, dil, upsample, pad_mode, in_relu, act, slope, epi, aux1, aux2, act2, out_scale, out,
                   ps=ps, aux_per_image=aux_per_image, bias_map=bias_map)
        if out_s is not None:                      # a preallocated Split (possibly a channel slice of a wider buffer)
            out_s.put(y.reshape(out_s.x.shape))
            sp = out_s
        else:
            sp = Split(y, out_h)
        return sp if out_split == "only" else (y, sp)
    h16 = False
    if isinstance(x, Split):
        assert DEFAULT_CONV_PRECISION in ("bf16x3", "f16") or precision == "bf16x3", "Split inputs need the bf16x3 / f16 mode"
        assert not in_relu and (x1 is None or (isinstance(x1, Split) and x1.h == x.h))
        h16 = x.h                                  # fp16 operands: the weights are rounded to fp16 as well (ops._f16_weights)
        x, x1 = x.x, (None if x1 is None else x1.x)
    else:
        assert not isinstance(x1, Split), "sources must both be split"
    x, N, H, W, C0, _ = _as_map(x)
    G = pc.groups
    if x1 is not None:
        x1, _, _, _, C1, _ = _as_map(x1)
        cat = torch.cat([x.reshape(N, H, W, G, C0 // G), x1.reshape(N, H, W, G, C1 // G)], -1).reshape(N, H, W, C0 + C1)
    else:
        cat = x
    assert cat.shape[-1] == pc.Cin
    inp = cat.permute(0, 3, 1, 2)
    if upsample:
        inp = F.interpolate(inp, scale_factor=2)
    if in_relu:
        inp = F.relu(inp)
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    if pad_mode == "replicate":
        inp = F.pad(inp, (pw, pw, ph, ph), mode="replicate")
        ph = pw = 0
    wd = _dense_weight(pc)
    # (Cout <= 4 layers read the fp16 map with fp32 weights and arithmetic: csrc/conv_direct.hip)
    y = F.conv2d(inp, f16_round(wd) if (h16 and pc.Cout // G > 4) else wd, None, stride, (ph, pw), dil, G)
    if pc.scale is not None:
        y = y * pc.scale.view(1, -1, 1, 1)
    if bias_map is not None:
        y = y + bias_map.reshape(y.shape[0], y.shape[2], y.shape[3], y.shape[1]).permute(0, 3, 1, 2)
    elif pc.bias is not None:
        y = y + pc.bias.view(1, -1, 1, 1)
    Nn, Co, Ho, Wo = y.shape
    y = y.permute(0, 2, 3, 1)
    tab = lambda a: a.reshape(1, Ho, Wo, Co) if aux_per_image else a.reshape(Nn, Ho, Wo, Co)
    if epi == "affine":                           # ABI 7: in front of the activation
        y = y * tab(aux2) + tab(aux1)
    elif epi == "ps_add2":
        assert ps is not None
        y = y + tab(aux1)
    if ps is not None:                            # column (ry, rx, c) of token cell (I, J) -> pixel (r*I + ry, r*J + rx), channel c
        r, c, g0, Hf, Wf = ps
        assert Co == g0 + (r - 1) * r * c and pc.groups == 1 and not out_nchw
        m = y.new_zeros(Nn, r * Ho, r * Wo, c)
        for ry in range(r):
            for rx in range(r):
                n0 = rx * c if ry == 0 else g0 + ((ry - 1) * r + rx) * c
                m[:, ry::r, rx::r] = y[..., n0:n0 + c]
        y = m[:, :Hf, :Wf]
        Ho, Wo, Co = Hf, Wf, c
        if epi == "ps_add2":
            y = y + aux2.reshape(Nn, Hf, Wf, c)
    y = _ACT[act](y, slope) * out_scale
    if epi == "mul":
        y = y * aux1.reshape(Nn, Ho, Wo, Co)
    elif epi == "add":
        y = _ACT[act2](y + aux1.reshape(Nn, Ho, Wo, Co), slope)
    elif epi == "gru":
        z, h = aux1.reshape(Nn, Ho, Wo, Co), aux2.reshape(Nn, Ho, Wo, Co)
        y = (1 - z) * h + z * y
    if out_nchw:
        res = y.permute(0, 3, 1, 2).contiguous()
        if out is not None:
            out.copy_(res)
            return out
        return res
    if out is None:
        return y.contiguous()
    out.copy_(y.reshape(out.shape))
    return out


def batched_gemm_nt(a, b, out, scale=1.0):
    """out[g] = scale * a[g] @ b[g]^T (fgt_conv_desc.gb_*: ops.batched_gemm_nt)."""
    out.copy_(torch.matmul(a.x, b.x.transpose(1, 2)) * scale)
    return out


def linear(x, pc, **kw):
    rows = x.shape[0]
    out = kw.pop("out", None)
    x1 = kw.pop("x1", None)
    img = lambda t: None if t is None else (t.view(1, 1, *t.shape) if isinstance(t, Split) else t.unsqueeze(0).unsqueeze(0))
    y = conv2d(img(x), pc, x1=img(x1), **kw)
    if isinstance(y, Split):                      # out_split = "only"
        return y.view(rows, pc.Cout)
    if isinstance(y, tuple):                      # out_split = "both"
        return y[0].reshape(rows, pc.Cout), y[1].view(rows, pc.Cout)
    y = y.reshape(rows, pc.Cout)
    if out is not None:
        out.copy_(y)
        return out
    return y


def _emit(val, out, as_split):
    """Write `val` into `out` (fp32 tensor or Split) if given, else return it (wrapped when as_split)."""
    if out is None:
        return Split(val) if as_split else val
    if isinstance(out, Split):
        out.put(val)
    else:
        out.copy_(val)
    return out


def layernorm(x0, gA, bA, x1=None, gB=None, bB=None, outA=None, outB=None, eps=1e-5, splitA=False, splitB=False):
    cat = x0 if x1 is None else torch.cat([x0, x1], 1)
    C = cat.shape[1]
    a = _emit(F.layer_norm(cat, (C,), gA, bA, eps), outA, splitA)
    if gB is None:
        return a
    return a, _emit(F.layer_norm(cat, (C,), gB, bB, eps), outB, splitB)


def _sdpa(q, k, v, h16=False):
    s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(q.size(-1))
    if not h16:
        return torch.matmul(F.softmax(s, dim=-1), v)
    # fp16 mode (csrc/attention_split.hip, H = true): the un-normalised probabilities exp(s - max) are rounded to fp16 for the PV
    # product, the normaliser is the fp32 sum of the unrounded ones (the kernel's online rescaling does the same per tile)
    p = torch.exp(s - s.amax(-1, keepdim=True))
    return torch.matmul(f16_round(p), v) / p.sum(-1, keepdim=True)


def _unsplit(*ts):
    """attention inputs: all fp32, or all Splits of one format (then only in the bf16x3 / f16 mode) -> (fp32 tensors, fp16 format?)"""
    if any(isinstance(t, Split) for t in ts):
        assert all(isinstance(t, Split) for t in ts) and DEFAULT_CONV_PRECISION in ("bf16x3", "f16"), "Split attention inputs need the bf16x3 / f16 mode, all together"
        assert all(t.h == ts[0].h for t in ts), "one Split format per call"
        return tuple(t.x for t in ts), ts[0].h
    return ts, False


def attention_temporal(qkv, b, t, nh, nw, heads, group, c, precision=None, out_split=False, tq=None):
    (qkv,), h16 = _unsplit(qkv)
    zh, zw, d = nh // group, nw // group, c // heads
    tq = t if tq is None else tq

    def zones(y, tt=t):
        return y.reshape(b, tt, group, zh, group, zw, heads, d).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(b, group * group, heads, -1, d)

    q = qkv[:, :c].reshape(b, t, nh * nw, c)[:, :tq].reshape(b * tq * nh * nw, c)        # queries: the first tq frames of each batch element
    a = _sdpa(zones(q, tq), zones(qkv[:, c:2 * c]), zones(qkv[:, 2 * c:3 * c]), h16)
    a = a.view(b, group, group, heads, tq, zh, zw, d).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(b * tq * nh * nw, c)
    return Split(a, h16) if out_split else a             # (an fp16 output needs fp16 inputs: ops._attn_out)


def _uncompact(m, bt, h, w, nh, nw, pad_row):
    """[bt*h*w (+ more), c] compact map -> [bt*nh*nw, c] on the padded grid, padded positions = row pad_row."""
    c = m.shape[1]
    out = m[pad_row].reshape(1, 1, 1, c).expand(bt, nh, nw, c).clone()
    out[:, :h, :w] = m[: bt * h * w].reshape(bt, h, w, c)
    return out.reshape(bt * nh * nw, c)


def attention_spatial(q, k, v, kg, vg, bt, h, w, nh, nw, heads, ws, n_global, precision=None, out_split=False, pad_row=None):
    (q, k, v, kg, vg), h16 = _unsplit(q, k, v, kg, vg)
    if pad_row is not None:
        q, k, v = (_uncompact(m, bt, h, w, nh, nw, pad_row) for m in (q, k, v))
    c = q.shape[1]
    gh, gw, d = nh // ws, nw // ws, c // heads

    def windows(y):
        return y.reshape(bt, gh, ws, gw, ws, c).transpose(2, 3).reshape(bt, gh * gw, ws * ws, c)

    def split(y):
        return y.reshape(bt, gh * gw, -1, heads, d).permute(0, 1, 3, 2, 4)

    K = torch.cat([windows(k), kg.reshape(bt, 1, n_global, c).expand(-1, gh * gw, -1, -1)], 2)
    V = torch.cat([windows(v), vg.reshape(bt, 1, n_global, c).expand(-1, gh * gw, -1, -1)], 2)
    a = _sdpa(split(windows(q)), split(K), split(V), h16)
    a = a.transpose(2, 3).reshape(bt, gh, gw, ws, ws, c).transpose(2, 3).reshape(bt, nh, nw, c)
    a = a[:, :h, :w].reshape(bt * h * w, c)
    return Split(a, h16) if out_split else a


def dw_pool(x0, x1, bt, nh, nw, k, w, bias, out, h=None, w_real=None):
    cat = x0 if x1 is None else torch.cat([x0, x1], 1)
    C = cat.shape[1]
    if h is not None and (h, w_real) != (nh, nw):          # compact maps: zero-pad to the window grid
        cat = F.pad(cat.reshape(bt, h, w_real, C), (0, 0, 0, nw - w_real, 0, nh - h)).reshape(bt * nh * nw, C)
    y = F.conv2d(cat.reshape(bt, nh, nw, C).permute(0, 3, 1, 2), w, bias, stride=k, groups=C)
    out.copy_(y.permute(0, 2, 3, 1).reshape(-1, C))
    return out


def dw3x3_residual(x, bt, h, w, wgt, bias):
    C = x.shape[-1]
    m = x.reshape(bt, h, w, C).permute(0, 3, 1, 2)
    return (F.conv2d(m, wgt, bias, 1, 1, 1, C) + m).permute(0, 2, 3, 1).contiguous().reshape(x.shape)


def fold(Y, frames, th, tw, Cc, k, s, p, Hf, Wf, normalize, res=None, out=None, relu=False, out_split=False):
    if isinstance(Y, Split):
        assert Y.h, "fold: a Split input must be the fp16 format"
        Y = Y.x
    cols = Y.reshape(frames, th * tw, k * k, Cc).permute(0, 3, 2, 1).reshape(frames, Cc * k * k, th * tw)  # back to (c, tap)
    f = F.fold(cols, (Hf, Wf), k, stride=s, padding=p)
    if normalize:
        f = f / F.fold(torch.ones(frames, k * k, th * tw), (Hf, Wf), k, stride=s, padding=p)
    f = f.permute(0, 2, 3, 1)
    if res is not None:
        f = res.reshape(frames, Hf, Wf, Cc) + f
    if relu:
        f = F.relu(f)
    return _emit(f.contiguous(), out, out_split)


def nchw_to_nhwc(src, dst, coff=0, zero_to=0, scale=1.0, shift=0.0):
    C = src.shape[1]
    dst[..., coff:coff + C] = src.permute(0, 2, 3, 1) * scale + shift
    if zero_to > C:
        dst[..., coff + C:coff + zero_to] = 0
    return dst


def nhwc_to_nchw(src):
    return _as_map(src)[0].permute(0, 3, 1, 2).contiguous()


def pad_tokens(src, bt, h, w, nh, nw, out=None):
    C = src.shape[1]
    m = src.reshape(bt, h, w, C)
    o = torch.zeros(bt, nh, nw, C)
    hh, ww = min(h, nh), min(w, nw)
    o[:, :hh, :ww] = m[:, :hh, :ww]
    o = o.reshape(bt * nh * nw, C)
    if out is not None:
        out.copy_(o)
        return out
    return o


def axpby(a, sa=1.0, b=None, sb=1.0, act=None, out=None, slope=0.2):
    v = a.reshape(-1, a.shape[-1]) * sa
    if b is not None:
        v = v + b * sb
    v = _ACT[act](v, slope)
    if out is not None:
        out.copy_(v)
        return out
    return v


# ------------------------------------------------------------------ flow-side kernels (spec)
def instnorm(x, act=None, res=None, act2=None, eps=1e-5, out=None, out_split=None, out_s=None):
    x4 = _as_map(x)[0]
    y = F.instance_norm(x4.permute(0, 3, 1, 2), eps=eps).permute(0, 2, 3, 1)
    y = _ACT[act](y, 0.2)
    if res is not None:
        y = _ACT[act2](y + res.reshape(y.shape), 0.2)
    y = y.contiguous()
    if out is not None:
        out.copy_(y)
        y = out
    if out_split:
        sp = Split(y, False) if out_s is None else out_s
        if out_s is not None:
            out_s.put(y.reshape(out_s.x.shape))
        return sp if out_split == "only" else (y, sp)
    return y


def avgpool2(src, rows, H, W):
    return F.avg_pool2d(src.reshape(rows, 1, H, W), 2, stride=2).reshape(rows, H // 2, W // 2)


def _sample(img, coords):
    H, W = img.shape[-2:]
    xg, yg = coords.split([1, 1], dim=-1)
    return F.grid_sample(img, torch.cat([2 * xg / (W - 1) - 1, 2 * yg / (H - 1) - 1], dim=-1), align_corners=True)


def laplace_fill(maps, masks, iters=1000, tol=1e-6, solver=None, bounds=None):
    """fgt_laplace_fill contract: the exact solution of the masked Laplace system (the kernel iterates to tol; the spec is the solve)."""
    import numpy as np
    from oracle import fill_oracle as FO
    n = masks.shape[0]
    out = [FO.regionfill(maps[b].numpy(), masks[b % n].numpy()) for b in range(maps.shape[0])]
    return torch.from_numpy(np.stack(out).astype(np.float32))


def corr_lookup(pyr, B, H1, W1, radius, coords, out=None, out_s=None):
    r = radius
    c = coords.reshape(B * H1 * W1, 1, 1, 2)
    d = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2)
    res = []
    for i, lvl in enumerate(pyr):
        Hl, Wl = H1 >> i, W1 >> i
        res.append(_sample(lvl.reshape(B * H1 * W1, 1, Hl, Wl), c / 2 ** i + delta).view(B, H1, W1, -1))
    val = torch.cat(res, -1)
    if out is not None:
        out.copy_(val)
    if out_s is not None:                                    # split output: the taps, then zero channels up to the tensor's width
        out_s.x.zero_()
        out_s.x[..., :val.shape[-1]].copy_(val)
    return out if out_s is None else (out_s if out is None else (out, out_s))


def convex_upsample(flow, mask):
    f4, B, H, W, _, _ = _as_map(flow)
    fl = f4[..., :2].permute(0, 3, 1, 2)
    mk = _as_map(mask)[0].permute(0, 3, 1, 2)
    mk = torch.softmax(mk.reshape(B, 1, 9, 8, 8, H, W), dim=2)
    up = F.unfold(8 * fl, [3, 3], padding=1).view(B, 2, 9, 1, 1, H, W)
    return torch.sum(mk * up, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(B, 2, 8 * H, 8 * W)


def gather_rows(src, ids, out=None):
    r = src[ids.long()]
    if out is not None:
        out.copy_(r)
        return out
    return r


def pack_frames(frames01, masks, ids=None, out=None):
    if ids is not None:
        frames01, masks = frames01[ids.long()], masks[ids.long()]
    r = torch.cat([(frames01 * 2 - 1) * (1 - masks), masks], 1).permute(0, 2, 3, 1).contiguous()
    if out is not None:
        out[..., :4].copy_(r)
        return out
    return r


def norm_flows(flows, n_out=None):
    n = flows.shape[-4]
    if n_out is not None and n_out > n:
        idx = list(range(n)) + [n - 1] * (n_out - n)
        flows = flows[..., idx, :, :, :]
    m = flows.flatten(-2).max(dim=-1, keepdim=True)[0]
    return flows / m.unsqueeze(-1)
