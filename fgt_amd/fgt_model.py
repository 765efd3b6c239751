"""MI355X-native FGT generator behind the reference's nn.Module API.

`Model(config).forward(masked_frames[b,t,3,H,W], flows[b,t,2,H,W], masks[b,t,1,H,W]) -> [b*t,3,H,W]`
has the constructor keys, forward signature and state_dict keys of FGT/models/model.py:12-25 (the
reference's `Model`), so `tool/video_inpainting.py:217-230,724` can import it unchanged.

The nn.Module tree below only *holds parameters* under the reference's names; no torch operator runs on
activations (tensor creation, views and slices only).  `forward` drives libfgt_hip.so (fgt_amd/ops.py): implicit-GEMM convolutions and
projections, flash attention with the zone/window gathers folded into addressing, fused LN / fold kernels.
"""
import math

import torch
import torch.nn as nn

import os

from . import ops
from .ops import PackedConv

# bf16x3 mode: q / k / v travel from the projection GEMMs to the attention as split tensors (FGT_SPLIT_ATTN=0: fp32 tensors re-split
# inside attn_bf16x3_kernel, the round-1 path, kept for A/B measurements)
SPLIT_ATTENTION = os.environ.get("FGT_SPLIT_ATTN", "1") != "0"


# bf16x3 mode: nn.Fold behind a per-token Linear (FusionFeedForward's first half, Vec2Patch) runs as ONE stride-1 3x3 convolution over the
# token grid with a sub-pixel epilogue (fgt_conv_desc.ps_r) on the tap-reusing kernel — no [tokens, 49*c] patch matrix, no fold pass.
# FGT_FOLD_CONV=0: Linear + fgt_fold (the fp32 / f16 modes' path), kept for A/B measurements.
FOLD_CONV = os.environ.get("FGT_FOLD_CONV", "1") != "0"
# the encoder's 8-group layer packed as 4 groups with block-diagonal weights (FGT._pack_encoder_layer); 0: as the reference groups it (A/B)
ENC_MERGE_GROUPS = os.environ.get("FGT_ENC_MERGE_GROUPS", "1") != "0"
# SWMHSA on three HIP streams when a call is too small to fill the chip (see FGT._spatial_attention): at most this many token rows per call
# (BASELINE config C2, t = 10: 7 200 rows; the benchmark's step batches 136 frames = 97 920 rows per call and stays on one stream).  0: never
SPATIAL_STREAM_ROWS = int(os.environ.get("FGT_SPATIAL_STREAM_ROWS", "32768"))
SPATIAL_STREAMS_EAGER = os.environ.get("FGT_SPATIAL_STREAMS_EAGER", "0") == "1"      # (tests / measurements: the three streams outside a capture too)
# tile order of the fold convolutions (fgt_conv_desc.tile_order): 1 = N-major inside an XCD (their 7 / 21 MB weight matrices do not fit the L2)
FOLD_TILE_ORDER = int(os.environ.get("FGT_FOLD_TILE_ORDER", "1"))


def fold_conv_supported(k, s, p):
    """fold(kernel k, stride s, padding p) of a per-token Linear is a 3x3 token-grid convolution with s x s sub-pixels when k = 2s + 1, p = s
    (the shipped 7 / 3 / 3, FGT/config/train.yaml:70-72): token offsets -1, 0, +1 cover every kernel position."""
    return k == 2 * s + 1 and p == s


def fold_conv_layout(cc, s):
    """Output columns of the fold convolution: [ry = 0: (rx, c) | zeros up to g0 | ry >= 1: (ry, rx, c)]; g0 is a tile boundary (128) so that
    the tiles of the ry >= 1 block skip the ky = 0 taps (fgt_conv_desc.ky_skip_n0).  Returns (g0, Cout, first column of sub-pixel (ry, rx))."""
    g0 = ops.ceil_to(s * cc, 128)
    col0 = lambda ry, rx: rx * cc if ry == 0 else g0 + ((ry - 1) * s + rx) * cc
    return g0, g0 + (s - 1) * s * cc, col0


def fold_conv_weight(w, cc, k, s):
    """Linear weight [cc*k*k, cin] (row c*k*k + ky*k + kx: the channel order of nn.Fold, ffn_base.py:39 / model.py:96) -> the conv weight
    [Cout, cin, 3, 3]: tap (a, b) = token offset (a - 1, b - 1) carries kernel position (s + ry - s*(a-1), s + rx - s*(b-1)) of sub-pixel
    (ry, rx) where it exists — pixel s*I + ry lies in the patch of token I + di at row ky = s*(I - (I + di)) + p + ry."""
    cin = w.shape[1]
    g0, cout, col0 = fold_conv_layout(cc, s)
    w4 = w.detach().float().view(cc, k, k, cin)
    W = torch.zeros(cout, cin, 3, 3, dtype=torch.float32, device=w.device)
    for ry in range(s):
        for rx in range(s):
            n0 = col0(ry, rx)
            for a in range(3):
                ky = s + ry - s * (a - 1)
                for b in range(3):
                    kx = s + rx - s * (b - 1)
                    if 0 <= ky < k and 0 <= kx < k:
                        W[n0:n0 + cc, :, a, b] = w4[:, ky, kx, :]
    return W


def fold_conv_tables(bias, cc, k, s, th, tw, normalize):
    """Per-position tables of the fold convolution on a th x tw token grid, [th*tw, Cout] fp32 on the CPU: `offset` = the Linear's biases summed
    over the (token, kernel position) pairs that reach the sub-pixel — tokens outside the grid contribute nothing, not even their bias — and
    `scale` = 1 / fold(ones) (ffn_base.py:58-66) or None; with normalisation the offset is pre-multiplied by the scale (epilogue: v*scale + offset)."""
    g0, cout, col0 = fold_conv_layout(cc, s)
    b4 = bias.detach().double().cpu().view(cc, k, k)

    def axis(n):        # valid[i, r, a]: token i + a - 1 exists and has kernel position s + r - s*(a-1); pos[r, a] = that position (clamped)
        i = torch.arange(n).view(n, 1, 1)
        r = torch.arange(s).view(1, s, 1)
        a = torch.arange(3).view(1, 1, 3)
        kp = s + r - s * (a - 1)
        ok = (kp >= 0) & (kp < k) & (i + a - 1 >= 0) & (i + a - 1 < n)
        return ok.double(), kp.clamp(0, k - 1).view(s, 3)

    vy, py = axis(th)
    vx, px = axis(tw)
    cnt = vy.sum(-1).view(th, 1, s, 1) * vx.sum(-1).view(1, tw, 1, s)                       # [th, tw, ry, rx]
    bsel = b4[:, py.view(s, 3, 1, 1), px.view(1, 1, s, 3)]                                  # [c, ry, a, rx, b]
    off = torch.einsum("iya,jxb,cyaxb->ijyxc", vy, vx, bsel)                                # [th, tw, ry, rx, c]
    scale = None
    if normalize:
        inv = 1.0 / cnt.clamp(min=1.0)
        off = off * inv.unsqueeze(-1)
        scale = torch.zeros(th, tw, cout, dtype=torch.float64)
    offset = torch.zeros(th, tw, cout, dtype=torch.float64)
    for ry in range(s):
        for rx in range(s):
            n0 = col0(ry, rx)
            offset[:, :, n0:n0 + cc] = off[:, :, ry, rx]
            if normalize:
                scale[:, :, n0:n0 + cc] = inv[:, :, ry, rx].unsqueeze(-1)
    f = lambda t: None if t is None else t.view(th * tw, cout).float().contiguous()
    return f(offset), f(scale)


# ----------------------------------------------------------------------------- parameter holders
class ConvParams(nn.Module):
    """weight [Cout, Cin/groups, kh, kw] + bias [Cout] (same names/shapes as nn.Conv2d)."""

    def __init__(self, cin, cout, k, groups=1, bias=True):
        super().__init__()
        kh, kw = (k, k) if isinstance(k, int) else k
        self.groups = groups
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, kh, kw))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        nn.init.normal_(self.weight, 0.0, 0.02)


class LinearParams(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.normal_(self.weight, 0.0, 0.02)


class LayerNormParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class Slot(nn.Module):
    """Parameter-free placeholder keeping the reference's Sequential/ModuleList indices (activations, pads)."""


class ConvBlockParams(nn.Module):
    """VanillaConv / GatedConv parameter names (FGT/models/utils/network_blocks_2d.py:7-95)."""

    def __init__(self, cin, cout, k, gated, bias=True):
        super().__init__()
        self.featureConv = ConvParams(cin, cout, k, bias=bias)
        if gated:
            self.gatingConv = ConvParams(cin, cout, k, bias=bias)


class DeconvBlockParams(nn.Module):
    def __init__(self, cin, cout, k, gated, bias=True):
        super().__init__()
        self.conv = ConvBlockParams(cin, cout, k, gated, bias)


class EncoderParams(nn.Module):
    """FGT/models/model.py:28-51."""
    SPEC = [(None, 64, 2, 1), (64, 64, 1, 1), (64, 128, 2, 1), (128, 256, 1, 1), (256, 384, 1, 1),
            (640, 512, 1, 2), (768, 384, 1, 4), (640, 256, 1, 8), (512, 128, 1, 1)]  # (cin, cout, stride, groups)

    def __init__(self, in_channels):
        super().__init__()
        mods = []
        for cin, cout, _, g in self.SPEC:
            mods += [ConvParams(cin or in_channels, cout, 3, groups=g), Slot()]
        self.layers = nn.ModuleList(mods)


class TMHSAParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.query_embedding = LinearParams(c, c)
        self.key_embedding = LinearParams(c, c)
        self.value_embedding = LinearParams(c, c)
        self.output_linear = LinearParams(c, c)


class SWMHSAParams(nn.Module):
    def __init__(self, c, cf, gd):
        super().__init__()
        self.query_embedding = LinearParams(c + cf, c)
        self.key_embedding = LinearParams(c + cf, c)
        self.value_embedding = LinearParams(c, c)
        self.output_linear = LinearParams(c, c)
        self.global_extract_v = ConvParams(c, c, gd, groups=c)
        self.global_extract_k = ConvParams(c + cf, c + cf, gd, groups=c + cf)
        self.q_norm = LayerNormParams(c + cf)
        self.k_norm = LayerNormParams(c + cf)
        self.v_norm = LayerNormParams(c)
        self.reweightFlow = nn.ModuleList([LinearParams(c + cf, cf), Slot()])


class FFNParams(nn.Module):
    """FusionFeedForward names: conv1, conv2.2 (ffn_base.py:39-45)."""

    def __init__(self, c, hidden):
        super().__init__()
        self.conv1 = LinearParams(c, hidden)
        self.conv2 = nn.ModuleList([Slot(), Slot(), LinearParams(hidden, c), Slot()])


class TemporalParams(nn.Module):
    def __init__(self, c, hidden):
        super().__init__()
        self.attention = TMHSAParams(c)
        self.ffn = FFNParams(c, hidden)
        self.norm1 = LayerNormParams(c)
        self.norm2 = LayerNormParams(c)


class SpatialParams(nn.Module):
    def __init__(self, c, cf, gd, hidden):
        super().__init__()
        self.attention = SWMHSAParams(c, cf, gd)
        self.ffn = FFNParams(c, hidden)
        self.norm = LayerNormParams(c)


class BlockParams(nn.Module):
    def __init__(self, c, cf, gd, hidden):
        super().__init__()
        self.t_transformer = TemporalParams(c, hidden)
        self.s_transformer = SpatialParams(c, cf, gd, hidden)


class PosEmbParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.proj = ConvParams(c, c, 3, groups=c)


class Vec2PatchParams(nn.Module):
    def __init__(self, c, cout):
        super().__init__()
        self.embedding = LinearParams(c, cout)


class DecoderParams(nn.Module):
    def __init__(self, c, gated, bias):
        super().__init__()
        self.layer1 = DeconvBlockParams(c, c, 3, gated, bias)
        self.layer2 = ConvBlockParams(c, c // 2, 3, gated, bias)
        self.layer3 = DeconvBlockParams(c // 2, c // 2, 3, gated, bias)
        self.final = ConvBlockParams(c // 2, 3, 3, gated, bias)


# ----------------------------------------------------------------------------- the network
class FGT(nn.Module):
    """Parameter tree of FGT/models/model.py:196-247 + the HIP forward of :249-283."""

    def __init__(self, t_groupSize, s_windowSize, g_downSize, input_resolution, in_channels, cnum, flow_inChannel,
                 flow_cnum, frame_hidden, flow_hidden, passmask, numBlocks, kernel_size, stride, padding, num_heads,
                 conv_type, norm, use_bias, ape, mlp_ratio=4, drop=0, init_weights=True):
        super().__init__()
        if conv_type not in ("vanilla", "gated"):
            raise NotImplementedError(f"conv_type={conv_type!r}: only 'vanilla' and 'gated' FGT variants are built "
                                      "(the reference's 'partial' blocks take (x, mask) tuples and cannot run in FGT.forward)")
        if norm not in (None, "None", "none"):
            raise NotImplementedError(f"norm={norm!r}: shipped FGT configs use norm='None' (FGT/config/train.yaml:76)")
        if drop != 0:
            raise NotImplementedError("dropout > 0 is training-only; inference path built with drop=0 (FGT/config/train.yaml:81)")
        if frame_hidden // num_heads != 128:
            raise NotImplementedError("attention kernels are built for head dim 128 (frame_hidden 512, 4 heads)")
        gated = conv_type == "gated"
        bias = bool(use_bias)
        self.cfg = dict(group=t_groupSize, ws=s_windowSize, gd=g_downSize, in_channels=in_channels, passmask=passmask,
                        heads=num_heads, k=tuple(kernel_size), s=tuple(stride), p=tuple(padding), ape=ape, gated=gated,
                        c=frame_hidden, cf=flow_hidden, cnum=cnum, mlp=mlp_ratio, flow_in=flow_inChannel)
        assert self.cfg["k"][0] == self.cfg["k"][1] and self.cfg["s"][0] == self.cfg["s"][1] and self.cfg["p"][0] == self.cfg["p"][1]
        self.in_channels = in_channels
        self.passmask = passmask
        self.ape = ape
        self.frame_endoder = EncoderParams(in_channels)  # (sic) reference key name, FGT/models/model.py:205
        self.flow_encoder = nn.ModuleList([
            Slot(),
            ConvBlockParams(flow_inChannel, flow_cnum, 5, gated, bias),
            ConvBlockParams(flow_cnum, flow_cnum * 2, 3, gated, bias),
            ConvBlockParams(flow_cnum * 2, flow_cnum * 2, 3, gated, bias),
            ConvBlockParams(flow_cnum * 2, flow_cnum * 2, 3, gated, bias)])
        self.patch2vec = ConvParams(cnum * 2, frame_hidden, kernel_size)
        self.f_patch2vec = ConvParams(flow_cnum * 2, flow_hidden, kernel_size)
        out_shape = (input_resolution[0] // 4, input_resolution[1] // 4)
        self.token_size = [int((out_shape[i] + 2 * padding[i] - kernel_size[i]) / stride[i] + 1) for i in range(2)]
        hidden = kernel_size[0] * kernel_size[1] * mlp_ratio
        if ape:
            self.add_pos_emb = PosEmbParams(frame_hidden)
        self.first_t_transformer = TemporalParams(frame_hidden, hidden)
        self.first_s_transformer = SpatialParams(frame_hidden, flow_hidden, g_downSize, hidden)
        self.transformer = nn.ModuleList([BlockParams(frame_hidden, flow_hidden, g_downSize, hidden)
                                          for _ in range(numBlocks // 2 - 1)])
        self.vec2patch = Vec2PatchParams(frame_hidden, kernel_size[0] * kernel_size[1] * cnum * 2)
        self.decoder = DecoderParams(cnum * 2, gated, bias)
        self._packed = None
        self._packed_key = None

    # ---- weight cache --------------------------------------------------------------------------
    def _cache_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def packed(self):
        key = self._cache_key()
        if self._packed is None or key != self._packed_key:
            self._packed = self._pack()
            self._packed_key = key
        return self._packed

    def prepack(self, dev=None):
        """Fill every lazily built cache of the forward pass on the CURRENT stream: the packed weights, their bf16 hi/lo (and, in the
        'f16' mode, fp16) images for the arithmetic mode selected now, and the zero rows of the spatial attention.  ClipRunner calls
        this before it forks the window groups onto side streams, so that no group reads a cache another stream is still writing."""
        P = self.packed()
        mode = ops.DEFAULT_CONV_PRECISION

        def walk(o):
            if isinstance(o, PackedConv):
                ops.prepack_weights(o, mode)
            elif isinstance(o, dict):
                for v in o.values():
                    walk(v)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    walk(v)
        walk(P)
        if mode == "bf16x3":                             # the decoder's two "nearest x2 + 3x3" layers run in their 2x2 sub-pixel form (ops.conv2d: _up4_ok)
            for blk in (P["dec"][0], P["dec"][2]):
                for o in blk:
                    if isinstance(o, PackedConv):
                        ops.prepack_up4(o)
        dev = dev if dev is not None else next(self.parameters()).device
        self._zero_row(dev, self.cfg["c"] + self.cfg["cf"])
        return self

    def _pack_encoder_layer(self, i, merge=True):
        """Encoder conv i (model.py:32-51).  The 640 -> 256 layer has 8 groups of 32 + 48 = 80 input and 32 output channels: 32-wide output tiles and
        a 48-channel second source that no 32-channel K-step fits (it ran at 92 TFLOP/s, a quarter of its neighbours).  ENC_MERGE_GROUPS (default)
        packs it as 4 groups of 64 + 96 -> 64 with block-diagonal weights: twice the multiply-adds, all of them with a zero weight, on 64-wide
        tiles whose K-steps are whole 32-channel chunks — so every layer of the encoder chain can hand its output over in the interleaved split
        layout.  The products that are not zero are the same ones; credited work stays the reference's (k_alg)."""
        cin, cout, _, g = EncoderParams.SPEC[i]
        conv = self.frame_endoder.layers[2 * i]
        w, b = conv.weight, conv.bias
        if not (merge and ENC_MERGE_GROUPS and g == 8 and cin is not None):
            return PackedConv(w, b, groups=g)
        c0 = EncoderParams.SPEC[3][1] // g                      # channels per group from the skip source x0 (the input of layer 4, model.py:58-59): 32
        c1 = w.shape[1] - c0                                    # ... from the previous layer: 48
        co = cout // g
        w = w.detach()
        wm = torch.zeros(cout, 2 * (c0 + c1), 3, 3, dtype=w.dtype, device=w.device)
        for gg in range(g):
            h, rows = gg % 2, slice(gg * co, (gg + 1) * co)
            wm[rows, h * c0:(h + 1) * c0] = w[rows, :c0]
            wm[rows, 2 * c0 + h * c1:2 * c0 + (h + 1) * c1] = w[rows, c0:]
        pc = PackedConv(wm, b, groups=g // 2)
        pc.k_alg = 9 * (c0 + c1)                                # the reference's K per output channel (the zero half is not work it does)
        return pc

    def _pack_block(self, blk):
        """ConvBlockParams -> (feature PackedConv, gating PackedConv | None)"""
        f = PackedConv(blk.featureConv.weight, blk.featureConv.bias)
        g = PackedConv(blk.gatingConv.weight, blk.gatingConv.bias) if self.cfg["gated"] else None
        return f, g

    def _pack_ffn(self, ffn):
        k2 = self.cfg["k"][0] * self.cfg["k"][1]
        w1, b1 = ffn.conv1.weight, ffn.conv1.bias
        cc = w1.shape[0] // k2
        # tap-major output columns: row (c*k2 + tap) -> (tap*cc + c)   (ffn_base.py:39 + Fold channel order)
        w1p = w1.detach().view(cc, k2, -1).permute(1, 0, 2).reshape(cc * k2, -1)
        b1p = b1.detach().view(cc, k2).permute(1, 0).reshape(-1)
        w2 = ffn.conv2[2].weight.detach()
        # Linear(hidden -> c) on unfold(...) == k x k / stride s conv over the folded map (ffn_base.py:40-45,56-75)
        w2c = w2.view(w2.shape[0], cc, self.cfg["k"][0], self.cfg["k"][1])
        P = dict(conv1=PackedConv(w1p, b1p), conv2=PackedConv(w2c, ffn.conv2[2].bias), cc=cc)
        P.update(self._pack_fold_conv(w1, b1, cc))
        return P

    def _pack_fold_conv(self, w, b, cc):
        """Linear + nn.Fold as a token-grid convolution (fold_conv_weight): the packed 3x3 weights, the bias on the CPU (the per-position tables
        depend on the token grid and are built on first use: _fold_tables) — only where the patch geometry allows it."""
        k, s, p = self.cfg["k"][0], self.cfg["s"][0], self.cfg["p"][0]
        if not (FOLD_CONV and fold_conv_supported(k, s, p) and cc % 4 == 0):
            return {}
        pc = PackedConv(fold_conv_weight(w, cc, k, s), None)
        pc.k_alg = w.shape[1]                                     # credited work: the Linear's (rows x cin x k*k*cc), not the zero taps
        return dict(fc=pc, fc_bias=b.detach().float().cpu(), fc_tables={})

    def _fold_tables(self, P, th, tw, dev, normalize):
        """(offset, scale) tables of a fold convolution for a th x tw token grid, on `dev` (built on the CPU, copied synchronously: usable from any
        stream afterwards)."""
        key = (th, tw, str(dev))
        if key not in P["fc_tables"]:
            off, sc = fold_conv_tables(P["fc_bias"], P["cc"], self.cfg["k"][0], self.cfg["s"][0], th, tw, normalize)
            P["fc_tables"][key] = (off.to(dev), None if sc is None else sc.to(dev))
        return P["fc_tables"][key]

    def _pack_temporal(self, m):
        a = m.attention
        wqkv = torch.cat([a.query_embedding.weight, a.key_embedding.weight, a.value_embedding.weight], 0)
        bqkv = torch.cat([a.query_embedding.bias, a.key_embedding.bias, a.value_embedding.bias], 0)
        return dict(qkv=PackedConv(wqkv, bqkv), out=PackedConv(a.output_linear.weight, a.output_linear.bias),
                    ffn=self._pack_ffn(m.ffn), n1=(m.norm1.weight.detach(), m.norm1.bias.detach()),
                    n2=(m.norm2.weight.detach(), m.norm2.bias.detach()))

    def _pack_spatial(self, m):
        a = m.attention
        f32 = lambda p: p.detach().float().contiguous()
        return dict(rw=PackedConv(a.reweightFlow[0].weight, a.reweightFlow[0].bias),
                    q=PackedConv(a.query_embedding.weight, a.query_embedding.bias),
                    k=PackedConv(a.key_embedding.weight, a.key_embedding.bias),
                    v=PackedConv(a.value_embedding.weight, a.value_embedding.bias),
                    out=PackedConv(a.output_linear.weight, a.output_linear.bias),
                    gk=(f32(a.global_extract_k.weight), f32(a.global_extract_k.bias)),
                    gv=(f32(a.global_extract_v.weight), f32(a.global_extract_v.bias)),
                    qn=(f32(a.q_norm.weight), f32(a.q_norm.bias)), kn=(f32(a.k_norm.weight), f32(a.k_norm.bias)),
                    vn=(f32(a.v_norm.weight), f32(a.v_norm.bias)),
                    ffn=self._pack_ffn(m.ffn), n=(f32(m.norm.weight), f32(m.norm.bias)))

    def _pack(self):
        P = {}
        P["enc"] = [self._pack_encoder_layer(i) for i in range(len(EncoderParams.SPEC))]
        # the reference's grouping of the 8-group layer for the modes whose chain is not the interleaved bf16x3 one (fp32-exact, f16): there the
        # merged packing only doubles that layer's multiply-adds
        P["enc_ref"] = [self._pack_encoder_layer(i, merge=False) if (ENC_MERGE_GROUPS and EncoderParams.SPEC[i][3] == 8) else pc for i, pc in enumerate(P["enc"])]
        P["fenc"] = [self._pack_block(self.flow_encoder[i]) for i in range(1, 5)]
        P["p2v"] = PackedConv(self.patch2vec.weight, self.patch2vec.bias)
        P["fp2v"] = PackedConv(self.f_patch2vec.weight, self.f_patch2vec.bias)
        if self.ape:
            P["pos"] = (self.add_pos_emb.proj.weight.detach().float().contiguous(),
                        self.add_pos_emb.proj.bias.detach().float().contiguous())
        P["t0"] = self._pack_temporal(self.first_t_transformer)
        P["s0"] = self._pack_spatial(self.first_s_transformer)
        P["blocks"] = [(self._pack_temporal(b.t_transformer), self._pack_spatial(b.s_transformer)) for b in self.transformer]
        k2 = self.cfg["k"][0] * self.cfg["k"][1]
        we, be = self.vec2patch.embedding.weight.detach(), self.vec2patch.embedding.bias.detach()
        cc = we.shape[0] // k2
        P["v2p"] = PackedConv(we.view(cc, k2, -1).permute(1, 0, 2).reshape(cc * k2, -1), be.view(cc, k2).permute(1, 0).reshape(-1))
        P["v2p_c"] = cc
        P["v2p_fc"] = dict(cc=cc, **self._pack_fold_conv(we, be, cc))
        d = self.decoder
        P["dec"] = [self._pack_block(d.layer1.conv), self._pack_block(d.layer2), self._pack_block(d.layer3.conv),
                    self._pack_block(d.final)]
        return P

    # ---- conv block helper (vanilla / gated) ---------------------------------------------------
    @staticmethod
    def _split_chain():
        """bf16x3 mode: conv -> conv chains hand their activations over pre-split (ops.Split): the producer's epilogue splits
        each value once and the consumer's loader is a plain LDS-DMA copy (csrc/conv_split.hip).  Same arithmetic as feeding
        fp32 tensors to the bf16x3 kernel, bit for bit."""
        return ops.DEFAULT_CONV_PRECISION in ("bf16x3", "f16")      # 'f16': the same chains hand over ONE fp16 plane (csrc/conv_f16.hip)

    @staticmethod
    def _f16():
        return ops.DEFAULT_CONV_PRECISION == "f16"

    def _block(self, x, packed, act="lrelu", out_split=None, **kw):
        f, g = packed
        if g is None:
            return ops.conv2d(x, f, act=act, out_split=out_split, **kw)
        gate = ops.conv2d(x, g, act="sigmoid", **kw)                      # network_blocks_2d.py:86-91
        return ops.conv2d(x, f, act=act, epi="mul", aux1=gate, out_split=out_split, **kw)

    # ---- transformer pieces --------------------------------------------------------------------
    def _ffn(self, y, x_res, P, bt, th, tw, Hf, Wf):
        """x_res + FusionFeedForward(y)  (ffn_base.py:53-77).  y may be a Split (bf16x3 mode)."""
        k, s, p = self.cfg["k"][0], self.cfg["s"][0], self.cfg["p"][0]
        sc = self._split_chain()
        if "fc" in P and sc and not self._f16() and isinstance(y, ops.Split):
            # Linear + fold / fold(ones) + ReLU as one 3x3 convolution over the token grid: K = 9 * c on the tap-reusing kernel, the folded map
            # written pre-split by the sub-pixel epilogue; the [tokens, k*k*cc] hidden matrix (768 MB per batched launch) never exists
            g0 = fold_conv_layout(P["cc"], s)[0]
            off, scale = self._fold_tables(P, th, tw, x_res.device, True)
            F = ops.conv2d(y.view(bt, th, tw, -1), P["fc"], stride=1, pad=1, act="relu", epi="affine", aux1=off, aux2=scale, aux_per_image=True,
                           ps=(s, P["cc"], g0, Hf, Wf), ky_skip_n0=g0, n_alg=k * k * P["cc"], out_split="only", out_il=ops.split_il(P["cc"]), tile_order=FOLD_TILE_ORDER)
            out = torch.empty_like(x_res)
            ops.conv2d(F, P["conv2"], stride=s, pad=p, epi="add", aux1=x_res, out=out.view(bt, th, tw, -1))
            return out
        # [bt*n, k*k*cc] tap-major; f16 mode: the hidden (the largest tensor of the block, 768 MB per batched launch as fp32) as fp16
        Y = ops.linear(y, P["conv1"], out_split="only" if self._f16() else None)
        # fold(x) / fold(ones); in split mode the ReLU in front of the second Linear is applied here, once per value,
        # and the map is handed over pre-split (otherwise it is applied on the gathered values inside the conv)
        F = ops.fold(Y, bt, th, tw, P["cc"], k, s, p, Hf, Wf, normalize=True, relu=sc, out_split=sc)
        out = torch.empty_like(x_res)
        ops.conv2d(F, P["conv2"], stride=s, pad=p, in_relu=not sc, epi="add", aux1=x_res, out=out.view(bt, th, tw, -1))
        return out

    def _temporal(self, x, P, b, t, th, tw, Hf, Wf, tq=None):
        """FGT/models/model.py:124-130 + attention_base.py:44-74.
        tq (clip scheduler, last temporal block): only the first tq frames of each of the b windows are produced — every frame still
        supplies keys and values, but queries, output projection and the FFN are per token, so the rows of the tq*b produced frames are
        the same values as in the full block; returns [b*tq*n, c]."""
        cfg = self.cfg
        G, c, bt = cfg["group"], cfg["c"], b * t
        zh, zw = math.ceil(th / G), math.ceil(tw / G)
        pad_r, pad_b = (zw - tw % zw) % zw, (zh - th % zh) % zh
        nh, nw = th + pad_b, tw + pad_r
        padded = bool(pad_r or pad_b)
        sc = self._split_chain()
        s = ops.layernorm(x, *P["n1"], splitA=sc and not padded)            # GEMM operands travel pre-split in bf16x3 mode
        if padded:
            s = ops.pad_tokens(s, bt, th, tw, nh, nw)
        # bf16x3 mode: the QKV GEMM hands q, k, v over pre-split; the attention streams K / V tiles by LDS-DMA (csrc/attention_split.hip)
        # (N-major tile order — fgt_conv_desc.tile_order — measured -3...5 % on this GEMM, 3.1 MB of weights, but it streams the 200 MB input once per
        #  N tile from beyond the L2: 2.9 GB of counter traffic per launch against 0.8 GB algorithmic; not used here)
        qkv = ops.linear(s, P["qkv"], out_split="only" if (sc and SPLIT_ATTENTION) else None)
        a = ops.attention_temporal(qkv, b, t, nh, nw, cfg["heads"], G, c, out_split=sc and not padded, tq=tq)
        if tq is not None and tq < t:
            n = th * tw
            x = x.view(b, t * n, -1)[:, : tq * n].reshape(b * tq * n, -1) if b > 1 else x[: tq * n]     # the residual rows of the produced frames
            bt = b * tq
        if padded:
            a = ops.pad_tokens(a, bt, nh, nw, th, tw)                      # crop (attention_base.py:71-72)
        x = ops.linear(a, P["out"], epi="add", aux1=x)
        y = ops.layernorm(x, *P["n2"], splitA=sc)
        return self._ffn(y, x, P["ffn"], bt, th, tw, Hf, Wf)

    def _spatial(self, x, f, P, bt, th, tw, Hf, Wf):
        """FGT/models/model.py:144-149: x + attention, then x + FFN(LN(x))."""
        x = self._spatial_attention(x, f, P, bt, th, tw)
        y = ops.layernorm(x, *P["n"], splitA=self._split_chain())
        return self._ffn(y, x, P["ffn"], bt, th, tw, Hf, Wf)

    def _spatial_attention(self, x, f, P, bt, th, tw):
        """x + SWMHSA(x, f)  (model.py:145 + attention_flow.py:57-113), global tokens projected once per frame.
        The reference zero-pads the token grid to a multiple of the window (attention_flow.py:120-124: 20x36 -> 24x40, a third more
        rows) and sends the padded tokens through the re-weighting, the LayerNorms and the q / k / v Linears; all of these act per
        token, so every padded token carries the SAME values: the projections of a zero row.  Those are computed once (row R of
        each map: one extra row of the same LayerNorm / GEMM launches) and the attention kernel reads that row for every padded
        position (fgt_attn_desc.compact) — the padded rows themselves never exist.  Bit-identical to the padded computation."""
        cfg = self.cfg
        ws, gd, c, cf = cfg["ws"], cfg["gd"], cfg["c"], cfg["cf"]
        pad_r, pad_b = (ws - tw % ws) % ws, (ws - th % ws) % ws
        nh, nw = th + pad_b, tw + pad_r
        R = bt * th * tw
        ng = (nh // gd) * (nw // gd)
        dev = x.device
        sc = self._split_chain()
        # (GEMM operands written by fgt_layernorm: interleaved hi/lo rows in bf16x3 mode — one 128-byte line per 32 channels for the wide kernel)
        new = (lambda r, ch: ops.Split.empty((r, ch), dev, interleaved=ops.split_il(ch))) if sc else (lambda r, ch: torch.empty(r, ch, dtype=torch.float32, device=dev))
        # LN outputs = GEMM operands: rows [0, R) real tokens, row R the padded token, rows R+1.. the frames' global tokens
        # Small calls (<= SPATIAL_STREAM_ROWS rows) keep their three operand buffers per shape with the padded token's rows written ONCE — the padded
        # token is a zero vector in front of the LayerNorms, so its rows are constants of the weights (LayerNorm(0) = beta): two of the module's 15
        # launches per call go, which is what a 7 200-row call is bound by.  Large calls allocate per call and write the rows with the two 1-row launches.
        # NEVER under a stream capture: a captured graph must own every buffer it touches.  (The first version cached under capture too, keyed by the stream:
        # torch hands out capture / side streams from a pool of 32, so a later capture could HIT an entry that an eager call had allocated on a recycled
        # stream id — the graph then replayed into memory the cache had long freed: a memory fault in the 400th test of the GPU suite, nowhere else.)
        pad_cached = x.is_cuda and 0 < R <= SPATIAL_STREAM_ROWS and not torch.cuda.is_current_stream_capturing()
        z = self._zero_row(dev, c + cf)
        pad_done = False
        if pad_cached:
            # (keyed by the calling stream too: ClipRunner may run window groups on concurrent streams, FGT_STREAMS > 1 — they must not share buffers)
            ck = (id(P), str(dev), torch.cuda.current_stream().cuda_stream, R, bt * ng, bool(sc), ops.mode_key())
            cache = self.__dict__.setdefault("_spatial_bufs", {})
            if ck not in cache:
                if len(cache) >= 32:
                    cache.clear()
                bufs = (new(R + 1 + bt * ng, c + cf), new(R + 1 + bt * ng, c), new(R + 1, c + cf))
                ops.layernorm(z[:, :c], *P["qn"], x1=z[:, c:], gB=P["kn"][0], bB=P["kn"][1], outA=bufs[2][R:R + 1], outB=bufs[0][R:R + 1])
                ops.layernorm(z[:, :c], *P["vn"], outA=bufs[1][R:R + 1])
                cache[ck] = (P, bufs)                # (P is held so that id(P) cannot be reused by another pack while the entry lives)
            kin, vin, q_ln = cache[ck][1]
            pad_done = True
        else:
            kin, vin, q_ln = new(R + 1 + bt * ng, c + cf), new(R + 1 + bt * ng, c), new(R + 1, c + cf)
        gk = torch.empty(bt * ng, c + cf, dtype=torch.float32, device=dev)
        gv = torch.empty(bt * ng, c, dtype=torch.float32, device=dev)
        osp = "only" if (sc and SPLIT_ATTENTION) else None
        # The module is a small dependency graph, not a chain: the value path (global-token pool of x, three LayerNorms, the v Linear) needs neither
        # the flow re-weighting nor the q / k path, and the k path parts from the q path behind the shared LayerNorm.  A call that cannot fill the
        # chip by itself (BASELINE config C2: 7 200 token rows = 224 tiles per 512-wide GEMM on 256 CUs x 2 workgroups) therefore runs on THREE
        # HIP streams — main: re-weighting -> LN(q|k) -> q -> attention -> out-projection; k: global-token pool of [x | f w] -> LN -> k;
        # v: the value path — so the under-filled launches share the chip (14 launches deep becomes 5).  Same kernels on the same data: bit-identical
        # to the single-stream order (tests/test_fgt_gpu.py); capturable (fork / join on the capturing stream).  Large calls (the clip runner's
        # 136-frame batches) fill the chip per launch and stay on one stream.
        # Only inside a stream capture: enqueued eagerly the module is bound by the host (15 launches + 6 cross-stream waits at 5-10 us each:
        # 0.38 ms on three streams against 0.30 on one, profiles/r06_run6_*), as a replayed graph by the GPU (0.26 -> 0.22 ms).
        par = x.is_cuda and 0 < R <= SPATIAL_STREAM_ROWS and (SPATIAL_STREAMS_EAGER or torch.cuda.is_current_stream_capturing())
        if par:
            main = torch.cuda.current_stream()
            s_k, s_v = self._side_streams(dev)
            s_v.wait_stream(main)
            ctx_v, ctx_k = torch.cuda.stream(s_v), torch.cuda.stream(s_k)
        else:
            import contextlib
            ctx_v = ctx_k = contextlib.nullcontext()
        with ctx_v:                                                                      # ---- value path (attention_flow.py:36, 94-96, 128-131)
            ops.dw_pool(x, None, bt, nh, nw, gd, *P["gv"], out=gv, h=th, w_real=tw)
            ops.layernorm(x, *P["vn"], outA=vin[:R])
            if not pad_done:
                ops.layernorm(z[:, :c], *P["vn"], outA=vin[R:R + 1])
            ops.layernorm(gv, *P["vn"], outA=vin[R + 1:])
            vv = ops.linear(vin, P["v"], out_split=osp)
        fw = ops.linear(x, P["rw"], x1=f, act="sigmoid", epi="mul", aux1=f)            # f * sigmoid(Linear([x|f]))  (attention_flow.py:52-55, 116-118)
        if par:
            s_k.wait_stream(main)
        with ctx_k:                                                                      # ---- global key tokens + the padded token's rows
            ops.dw_pool(x, fw, bt, nh, nw, gd, *P["gk"], out=gk, h=th, w_real=tw)
            ops.layernorm(gk, *P["kn"], outA=kin[R + 1:])
            if not pad_done:
                ops.layernorm(z[:, :c], *P["qn"], x1=z[:, c:], gB=P["kn"][0], bB=P["kn"][1], outA=q_ln[R:R + 1], outB=kin[R:R + 1])
        ops.layernorm(x, *P["qn"], x1=fw, gB=P["kn"][0], bB=P["kn"][1], outA=q_ln[:R], outB=kin[:R])
        if par:
            s_k.wait_stream(main)                                                        # kin[:R] is written on the main stream
            main.wait_stream(s_k)                                                        # q_ln[R] on the k stream
        with ctx_k:
            kk = ops.linear(kin, P["k"], out_split=osp)
        q = ops.linear(q_ln, P["q"], out_split=osp)
        if par:
            main.wait_stream(s_k)
            main.wait_stream(s_v)
        a = ops.attention_spatial(q, kk[:R + 1], vv[:R + 1], kk[R + 1:], vv[R + 1:], bt, th, tw, nh, nw, cfg["heads"], ws, ng, out_split=sc,
                                  pad_row=R)
        return ops.linear(a, P["out"], epi="add", aux1=x)

    def _side_streams(self, dev):
        cache = self.__dict__.setdefault("_streams", {})
        key = str(dev)
        if key not in cache:
            cache[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        return cache[key]

    def _zero_row(self, dev, ch):
        """One row of zeros (the reference's padded token before the LayerNorms), kept per device."""
        cache = self.__dict__.setdefault("_zero_rows", {})
        key = (str(dev), ch)
        if key not in cache:
            cache[key] = torch.zeros(1, ch, dtype=torch.float32, device=dev)
        return cache[key]

    # ---- per-frame stages (exposed separately so the clip scheduler can cache them) -------------
    def token_grid(self, H, W):
        """(th, tw) of an H x W input: soft split of the H/4 x W/4 feature map (model.py:218-228 for the configured resolution)."""
        cfg = self.cfg
        return tuple((n // 4 + 2 * cfg["p"][i] - cfg["k"][i]) // cfg["s"][i] + 1 for i, n in enumerate((H, W)))

    def encode_frames(self, masked_frames=None, flows=None, masks=None, packed_in=None, out=None):
        """model.py:253-262: returns (enc_feats [bt,Hf,Wf,2cnum], tokens [bt*n,c], flow tokens [bt*n,cf], th, tw).
        packed_in = (x_in [bt,H,W,4], f_in [bt,H,W,4]): channels-last inputs already packed on device (ops.pack_frames /
        ops.nchw_to_nhwc; the clip scheduler's path) instead of the three NCHW tensors of the nn.Module API;
        out = (enc, tok, ftok): preallocated destinations (slices of the clip's feature buffers) written in place."""
        P = self.packed()
        cfg = self.cfg
        if packed_in is None:
            b, t, _, H, W = masked_frames.shape
            if H % 4 or W % 4:
                raise ValueError(f"FGT needs H, W divisible by 4, got {H}x{W}")
            bt = b * t
            dev = masked_frames.device
            cin = ops.ceil_to(self.in_channels, 4)
            x_in = torch.empty(bt, H, W, cin, dtype=torch.float32, device=dev)
            ops.nchw_to_nhwc(masked_frames.reshape(bt, 3, H, W).float(), x_in, coff=0,
                             zero_to=cin if not self.passmask else 0)
            if self.passmask:
                ops.nchw_to_nhwc(masks.reshape(bt, 1, H, W).float(), x_in, coff=3, zero_to=cin - 3)
            fin = ops.ceil_to(cfg["flow_in"], 4)
            f_in = torch.empty(bt, H, W, fin, dtype=torch.float32, device=dev)
            ops.nchw_to_nhwc(flows.reshape(bt, cfg["flow_in"], H, W).float(), f_in, coff=0, zero_to=fin)
        else:
            x_in, f_in = packed_in
            bt, H, W, _ = x_in.shape
            assert self.passmask and self.in_channels == 4, "packed inputs carry (masked frame | mask)"
        th, tw = self.token_grid(H, W)
        o_enc, o_tok, o_ftok = out if out is not None else (None, None, None)
        merged = ENC_MERGE_GROUPS and ops.DEFAULT_CONV_PRECISION == "bf16x3"
        E = P["enc"] if merged else P["enc_ref"]
        strides = [sp[2] for sp in EncoderParams.SPEC]
        sc = "only" if self._split_chain() else None
        e = x_in
        x0 = None
        for i in range(9):
            if i == 4:
                x0 = e                                                         # model.py:58-59
            osp = ("both" if sc else None) if i == 8 else sc                   # the last layer also feeds fold()'s fp32 residual
            dst = o_enc if i == 8 else None
            # split outputs: interleaved wherever the consumer's K-steps are whole 32-channel chunks of each source — every layer with the 8-group
            # layer merged into 4 groups (bf16x3 chain); with the reference's grouping (48 channels per group from the second source) layers 3-7
            # hand over planes, since both sources of a conv share one layout
            il = ops.split_il(E[i].Cout) and (i <= 2 or i == 8 or merged)
            if i <= 4:
                e = ops.conv2d(e, E[i], stride=strides[i], pad=1, act="lrelu", out_split=osp, out=dst, out_il=il)
            else:
                e = ops.conv2d(x0, E[i], x1=e, stride=1, pad=1, act="lrelu", out_split=osp, out=dst, out_il=il)    # grouped concat, model.py:60-65
        enc, enc_in = e if sc else (e, e)
        FE = P["fenc"]
        il = lambda blk: ops.split_il(blk[0].Cout)
        fe = self._block(f_in, FE[0], stride=1, pad=2, pad_mode="replicate", out_split=sc, out_il=il(FE[0]))    # ReplicationPad2d(2) + 5x5 conv
        fe = self._block(fe, FE[1], stride=2, pad=1, out_split=sc, out_il=il(FE[1]))
        fe = self._block(fe, FE[2], stride=1, pad=1, out_split=sc, out_il=il(FE[2]))
        fe = self._block(fe, FE[3], stride=2, pad=1, out_split=sc, out_il=il(FE[3]))
        s, p = cfg["s"][0], cfg["p"][0]
        tok = ops.conv2d(enc_in, P["p2v"], stride=s, pad=p, out=None if o_tok is None else o_tok.view(bt, th, tw, -1))
        ftok = ops.conv2d(fe, P["fp2v"], stride=s, pad=p, out=None if o_ftok is None else o_ftok.view(bt, th, tw, -1))
        assert (tok.shape[1], tok.shape[2]) == (th, tw)
        return enc, tok.view(bt * th * tw, -1), ftok.view(bt * th * tw, -1), th, tw

    def transform_decode(self, enc, x, f, b, t, th, tw, n_out=None, keep=None, tq=None, keep_q=None):
        """model.py:272-283 given per-frame features.  Soft composition + decoder run only for the frames whose output is
        consumed — the tool discards the decoded reference frames (tool/video_inpainting.py:727: only
        `range(len(neighbor_ids))` is read) and both stages are per-frame, so the kept frames are unchanged:
        `n_out` (b == 1): the first n_out frames; `keep` (any b): an int32 device tensor of frame indices in [0, b*t).
        `tq` (with keep / n_out): the consumed frames are among the first tq of each window's t frames.  Then the LAST transformer pair is
        pruned as well: its temporal block produces only those tq frames per window (all t frames still act as keys / values) and its
        spatial block — per frame throughout — runs on them alone; everything the kept frames depend on is computed exactly as before.
        `keep_q`: `keep` re-indexed for the pruned layout ((j*t + i) -> j*tq + i), precomputed by the caller (else derived here)."""
        P = self.packed()
        cfg = self.cfg
        bt = b * t
        n = th * tw
        Hf, Wf = enc.shape[1], enc.shape[2]
        if keep is not None and keep.dtype != torch.int32:
            keep = keep.to(torch.int32)                                 # (the row gathers take int32 indices; int64 was the documented type)
        if keep_q is not None and keep_q.dtype != torch.int32:
            keep_q = keep_q.to(torch.int32)
        if n_out is not None and tq is None:
            tq = n_out                                                  # b == 1: the consumed frames are the prefix itself
        prune = tq is not None and 0 < tq < t and (keep is not None or n_out is not None) and len(P["blocks"]) > 0
        x = self._temporal(x, P["t0"], b, t, th, tw, Hf, Wf)
        if self.ape:
            x = ops.dw3x3_residual(x.view(bt, th, tw, -1), bt, th, tw, *P["pos"]).view(bt * th * tw, -1)
        x = self._spatial(x, f, P["s0"], bt, th, tw, Hf, Wf)
        for i, (pt, ps) in enumerate(P["blocks"]):
            if prune and i == len(P["blocks"]) - 1:
                x = self._temporal(x, pt, b, t, th, tw, Hf, Wf, tq=tq)                   # [b*tq*n, c]
                fq = f.view(b, t * n, -1)[:, : tq * n].reshape(b * tq * n, -1) if b > 1 else f[: tq * n]
                x = self._spatial(x, fq, ps, b * tq, th, tw, Hf, Wf)
            else:
                x = self._temporal(x, pt, b, t, th, tw, Hf, Wf)
                x = self._spatial(x, f, ps, bt, th, tw, Hf, Wf)
        if keep is not None:
            assert n_out is None
            if prune:                                                   # frame j*t + i of the full layout is frame j*tq + i of the pruned one
                enc = ops.gather_rows(enc, keep)
                keep = keep_q if keep_q is not None else (torch.div(keep, t, rounding_mode="floor") * tq + keep % t).to(torch.int32)
                x = ops.gather_rows(x.view(b * tq, n, -1), keep).view(keep.numel() * n, -1)
            else:
                x = ops.gather_rows(x.view(bt, n, -1), keep).view(keep.numel() * n, -1)
                enc = ops.gather_rows(enc, keep)
            bt = keep.numel()
        elif n_out is not None and n_out < bt:
            assert b == 1, "n_out needs a single clip (frames of one batch element are contiguous); use keep= for b > 1"
            bt = n_out
            x, enc = x[: bt * n], enc[:bt]
        sc = "only" if self._split_chain() else None
        VP = P["v2p_fc"]
        if "fc" in VP and sc and not self._f16():
            # Vec2Patch (Linear 512 -> 49*128 + nn.Fold, model.py:102-110) + the encoder residual (model.py:280) as one 3x3 token-grid convolution
            k, s = cfg["k"][0], cfg["s"][0]
            g0 = fold_conv_layout(VP["cc"], s)[0]
            off, _ = self._fold_tables(VP, th, tw, x.device, False)
            xs = ops.split(x, interleave=ops.split_il(x.shape[1]))
            feat = ops.conv2d(xs.view(bt, th, tw, -1), VP["fc"], stride=1, pad=1, epi="ps_add2", aux1=off, aux2=enc, aux_per_image=True,
                              ps=(s, VP["cc"], g0, Hf, Wf), ky_skip_n0=g0, n_alg=k * k * VP["cc"], out_split="only", out_il=ops.split_il(VP["cc"]), tile_order=FOLD_TILE_ORDER)
        else:
            if self._f16():
                # the token stream is fp32; rounding it once here (one pass over [rows, 512]) lets the widest GEMM of the path (512 -> 6272) run
                # on the fp16 kernel and hand fold() an fp16 patch matrix (1.6 GB per clip pass as fp32)
                Y = ops.linear(ops.split(x), P["v2p"], out_split="only")
            else:
                Y = ops.linear(x, P["v2p"])
            # soft composition + encoder residual; split mode: written pre-split for the decoder's first conv (its only consumer)
            feat = ops.fold(Y, bt, th, tw, P["v2p_c"], cfg["k"][0], cfg["s"][0], cfg["p"][0], Hf, Wf, normalize=False, res=enc, out_split=bool(sc))
        D = P["dec"]
        y = self._block(feat, D[0], stride=1, pad=1, upsample=True, out_split=sc, out_il=ops.split_il(D[0][0].Cout))
        y = self._block(y, D[1], stride=1, pad=1, out_split=sc, out_il=ops.split_il(D[1][0].Cout))
        # the Cout = 3 kernel below gathers fp32 — or, in the f16 mode, the fp16 map (64 channels at full resolution: the largest activation
        # of the path, 2.3 GB per clip pass as fp32); its LDS-tiled form needs an 8 x 32 output tile
        h_last = self._f16() and 4 * Hf >= 8 and 4 * Wf >= 32 and D[3][0].Cg % 16 == 0 and D[3][0].Cout <= 4
        y = self._block(y, D[2], stride=1, pad=1, upsample=True, out_split="only" if h_last else None)
        if D[3][1] is None:
            return ops.conv2d(y, D[3][0], stride=1, pad=1, act="tanh", out_nchw=True)   # final conv + torch.tanh
        y = self._block(y, D[3], act=None, stride=1, pad=1)
        return ops.nhwc_to_nchw(ops.axpby(y, act="tanh").view(y.shape))

    def forward(self, masked_frames, flows, masks):
        b, t = masked_frames.shape[:2]
        with torch.no_grad():
            enc, x, f, th, tw = self.encode_frames(masked_frames, flows, masks)
            return self.transform_decode(enc, x, f, b, t, th, tw)


class Model(nn.Module):
    """Drop-in for FGT.models.model.Model (FGT/models/model.py:12-25)."""

    def __init__(self, config):
        super().__init__()
        self.net = FGT(config['tw'], config['sw'], config['gd'], config['input_resolution'], config['in_channel'],
                       config['cnum'], config['flow_inChannel'], config['flow_cnum'], config['frame_hidden'],
                       config['flow_hidden'], config['PASSMASK'], config['numBlocks'], config['kernel_size'],
                       config['stride'], config['padding'], config['num_head'], config['conv_type'], config['norm'],
                       config['use_bias'], config['ape'], config['mlp_ratio'], config['drop'], config['init_weights'])

    def forward(self, frames, flows, masks):
        return self.net(frames, flows, masks)


DEFAULT_CONFIG = dict(tw=2, sw=8, gd=4, input_resolution=(240, 432), in_channel=4, cnum=64, flow_inChannel=2,
                      flow_cnum=64, frame_hidden=512, flow_hidden=256, PASSMASK=1, numBlocks=8, kernel_size=(7, 7),
                      stride=(3, 3), padding=(3, 3), num_head=4, conv_type='vanilla', norm='None', use_bias=1, ape=1,
                      mlp_ratio=40, drop=0, init_weights=1)   # FGT/config/train.yaml:59-85 + FGT/inputs.py:48
