"""MI355X-native LAFC flow-completion network behind the reference's nn.Module API.

`Model(config).forward(flows[b,2,T,H,W], masks[b,1,T,H,W], edges=None) -> (flow[b,2,H,W], edge[b,1,H,W])`
with the constructor keys and state_dict keys of LAFC/models/lafc.py:6-15 (`tool/video_inpainting.py:200-214,378`).

The T frames of a clip window are a batch of channels-last images: a (1,k,k) Conv3d is the 2-D implicit-GEMM conv over
T images, a (3,1,1) Conv3d is the same kernel run with "height" = T and "width" = H*W (kernel 3x1), so the whole P3D
net runs on fgt_conv2d with fused LeakyReLU / residual / nearest-x2 / skip-concat (two-source) epilogues.
"""
import os

import torch
import torch.nn as nn

from . import ops
from .fgt_model import ConvParams, Slot
from .ops import PackedConv


class Conv3dParams(nn.Module):
    def __init__(self, cin, cout, k, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, *k))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        nn.init.kaiming_normal_(self.weight, a=0, mode="fan_in")


class Block3d(nn.Module):
    def __init__(self, cin, cout, k, gated, bias):
        super().__init__()
        self.featureConv = Conv3dParams(cin, cout, k, bias)
        if gated:
            self.gatingConv = Conv3dParams(cin, cout, k, bias)


class Block2d(nn.Module):
    def __init__(self, cin, cout, k, gated, bias=True):
        super().__init__()
        self.featureConv = ConvParams(cin, cout, k, bias=bias)
        nn.init.kaiming_normal_(self.featureConv.weight, a=0, mode="fan_in")
        if gated:
            self.gatingConv = ConvParams(cin, cout, k, bias=bias)
            nn.init.kaiming_normal_(self.gatingConv.weight, a=0, mode="fan_in")


class Deconv2d(nn.Module):
    def __init__(self, cin, cout, k, gated, bias=True):
        super().__init__()
        self.conv = Block2d(cin, cout, k, gated, bias)


class P3DParams(nn.Module):
    def __init__(self, cin, cout, k, stride, pad, gated, bias, residual):
        super().__init__()
        self.conv1 = Block3d(cin, cout, (1, k, k), gated, bias)
        self.conv2 = Block3d(cout, cout, (3, 1, 1), gated, bias)
        self.geom = (k, stride, pad, residual)


class EdgeParams(nn.Module):
    def __init__(self, gated):
        super().__init__()
        self.projection = Block2d(2, 16, 3, gated)
        self.mid_layer_1 = Block2d(16, 16, 3, gated)
        self.mid_layer_2 = Block2d(16, 16, 3, gated)
        self.out_layer = Block2d(16, 1, 1, gated)


class P3DNet(nn.Module):
    """Parameter tree of LAFC/models/lafc.py:18-82, HIP forward of :84-105."""

    def __init__(self, num_flows, num_feats, in_channels, passmask, use_residual, res_blocks, use_bias, conv_type, init_weights):
        super().__init__()
        if conv_type not in ("vanilla", "gated"):
            raise NotImplementedError(f"conv_type={conv_type!r}: the reference's 3-D PartialConv is broken "
                                      "(LAFC/models/utils/network_blocks.py:116) and never used")
        g, b, nf = conv_type == "gated", bool(use_bias), num_feats
        self.gated, self.passmask, self.num_flows, self.in_channels, self.resNums = g, passmask, num_flows, in_channels, res_blocks
        self.encoder2 = nn.ModuleList([Slot(), P3DParams(in_channels, nf, 5, 1, 0, g, b, 0), P3DParams(nf, nf * 2, 3, 2, 1, g, b, 0)])
        self.encoder4 = nn.ModuleList([P3DParams(nf * 2, nf * 2, 3, 1, 1, g, b, use_residual), P3DParams(nf * 2, nf * 4, 3, 2, 1, g, b, 0)])
        shared = P3DParams(nf * 4, nf * 4, 3, 1, 1, g, b, 1)
        self.res_blocks = nn.ModuleList([shared for _ in range(res_blocks)])       # one instance reused (lafc.py:38-43)
        self.condense2 = Block3d(nf * 2, nf * 2, (num_flows, 1, 1), g, b)
        self.condense4_pre = Block3d(nf * 4, nf * 4, (num_flows, 1, 1), g, b)
        self.condense4_post = Block3d(nf * 4, nf * 4, (num_flows, 1, 1), g, b)
        self.middle = nn.ModuleList([Block2d(nf * 4, nf * 4, 3, g, b) for _ in range(4)])
        self.decoder2 = nn.ModuleList([Deconv2d(nf * 8, nf * 2, 3, g, b), Block2d(nf * 2, nf * 2, 3, g, b), Block2d(nf * 2, nf * 2, 3, g, b)])
        self.decoder = nn.ModuleList([Deconv2d(nf * 4, nf, 3, g, b), Block2d(nf, nf // 2, 3, g, b), Block2d(nf // 2, 2, 3, g, b)])
        self.edgeDetector = EdgeParams(g)
        self._packed, self._key = None, None
        self.use_graph = os.environ.get("FGT_GRAPHS", "0") == "1"
        self._graphs = None

    # ---- packing
    def _pk(self, blk, temporal=False):
        def one(cp):
            w = cp.weight.detach()
            if w.dim() == 5:   # Conv3d: (1,k,k) -> [Cout,Cin,k,k]; (T,1,1) -> [Cout,Cin,T,1]
                w = w[:, :, 0] if w.shape[2] == 1 and not temporal else w[:, :, :, 0]
            return PackedConv(w, None if cp.bias is None else cp.bias)
        return one(blk.featureConv), (one(blk.gatingConv) if self.gated else None)

    def packed(self):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is None or key != self._key:
            P = {}
            for name, mods in (("encoder2", self.encoder2[1:]), ("encoder4", self.encoder4), ("res", self.res_blocks)):
                P[name] = [(self._pk(m.conv1), self._pk(m.conv2, temporal=True), m.geom) for m in mods]
            for name in ("condense2", "condense4_pre", "condense4_post"):
                P[name] = self._pk(getattr(self, name), temporal=True)
            P["middle"] = [self._pk(m) for m in self.middle]
            P["decoder2"] = [self._pk(self.decoder2[0].conv), self._pk(self.decoder2[1]), self._pk(self.decoder2[2])]
            P["decoder"] = [self._pk(self.decoder[0].conv), self._pk(self.decoder[1]), self._pk(self.decoder[2])]
            e = self.edgeDetector
            P["edge"] = [self._pk(e.projection), self._pk(e.mid_layer_1), self._pk(e.mid_layer_2), self._pk(e.out_layer)]
            self._packed, self._key = P, key
        return self._packed

    def _block(self, x, packed, act="lrelu", epi=None, aux1=None, act2=None, slope=0.2, out=None, **kw):
        """Vanilla: one fused conv.  Gated (network_blocks.py:59-91): sigmoid(gatingConv) * act(featureConv)."""
        f, g = packed
        if g is None:
            return ops.conv2d(x, f, act=act, slope=slope, epi=epi, aux1=aux1, act2=act2, out=out, **kw)
        gate = ops.conv2d(x, g, act="sigmoid", **kw)
        if epi is None:
            return ops.conv2d(x, f, act=act, slope=slope, epi="mul", aux1=gate, out=out, **kw)
        assert epi == "add" and (out is None or out.is_contiguous())
        y = ops.conv2d(x, f, act=act, slope=slope, epi="mul", aux1=gate, **kw)
        C = y.shape[-1]
        tgt = torch.empty_like(y) if out is None else out
        ops.axpby(y.reshape(-1, C), 1.0, aux1.reshape(-1, C), 1.0, act=act2, slope=slope, out=tgt.reshape(-1, C))
        return tgt

    # ---- split chains (vanilla convs, bf16x3 arithmetic): a conv whose output only feeds other convs writes it PRE-SPLIT (ops.Split, bf16 hi /
    # lo pair) and the consumer's im2col tiles become plain LDS-DMA copies (csrc/conv_split.hip) — the same products as handing fp32 tensors to
    # the bf16x3 kernel (tap-routed layers sum them in another order, csrc/conv_taps.hip: equal to fp32 rounding, not bit for bit).  `want`: "s" = split only, "both" = (fp32, split) for outputs that are ALSO an epilogue operand (residuals) or
    # feed a Cout <= 4 VALU conv, "f32" = fp32 only.  Without split chains (exact-fp32 mode, gated convs) every form is the fp32 tensor.
    def _sc(self):
        return (not self.gated) and ops.DEFAULT_CONV_PRECISION != "fp32"

    def _cv(self, x, packed, want="s", **kw):
        """One conv block; returns (tensor for conv consumers, fp32 tensor or None)."""
        if not self._sc():
            y = self._block(x, packed, **kw)
            return y, y
        if want == "f32":
            y = self._block(x, packed, **kw)
            return y, y
        r = ops.conv2d(x, packed[0], act=kw.pop("act", "lrelu"), slope=kw.pop("slope", 0.2), out_split="only" if want == "s" else "both", out_h=False, **kw)
        return (r, None) if want == "s" else (r[1], r[0])

    @staticmethod
    def _v(t, *shape):
        return t.view(*shape)

    def _p3d(self, x, b, T, blk, want="s", x_f32=None):
        """x [b*T, H, W, C] (fp32 tensor or Split) -> P3DBlock output (lafc.py:117-125) in the forms `want` asks for;
        x_f32: the fp32 form of x (the residual operand) when x is a Split."""
        c1, c2, (k, stride, pad, residual) = blk
        y, _ = self._cv(x, c1, "s", stride=stride, pad=pad if k != 5 else 2, pad_mode="zeros" if k != 5 else "replicate")
        bT, H, W, C = y.shape
        yt = y.view(b, T, H * W, C)
        if residual:
            res = (x_f32 if x_f32 is not None else x).view(b, T, H * W, C)
            o, o32 = self._cv(yt, c2, want, pad=(1, 0), epi="add", aux1=res)
        else:
            o, o32 = self._cv(yt, c2, want, pad=(1, 0))
        return o.view(bT, H, W, C), (None if o32 is None else o32.view(bT, H, W, C))

    def _condense(self, x, b, T, pk, want="s"):
        bT, H, W, C = x.shape
        o, o32 = self._cv(x.view(b, T, H * W, C), pk, want, pad=0)
        return o.view(b, H, W, -1)

    def forward(self, flows, masks, edges=None):
        with torch.no_grad():
            if self.use_graph and edges is None and flows.is_cuda:
                from .graph import GraphCache
                if self._graphs is None:
                    self._graphs = GraphCache(lambda a, b: self._forward(a, b, None),
                                              state_key=lambda: tuple((p.data_ptr(), p._version) for p in self.parameters()))
                return tuple(o.clone() for o in self._graphs(flows.float(), masks.float()))
            return self._forward(flows, masks, edges)

    def _forward(self, flows, masks, edges):
        P = self.packed()
        b, _, T, H, W = flows.shape
        if T != self.num_flows:
            raise ValueError(f"LAFC was built for {self.num_flows} flows per call, got {T}")
        if H % 4 or W % 4:
            raise ValueError("LAFC needs H, W divisible by 4")
        dev = flows.device
        cin = ops.ceil_to(self.in_channels, 4)
        x = torch.zeros(b * T, H, W, cin, dtype=torch.float32, device=dev)
        parts = [flows] + ([masks] if self.passmask else []) + ([edges] if edges is not None else [])
        off = 0
        for p in parts:
            c = p.shape[1]
            ops.nchw_to_nhwc(p.permute(0, 2, 1, 3, 4).reshape(b * T, c, H, W).float(), x, coff=off)
            off += c
        assert off == self.in_channels, f"input channels {off} != in_channel {self.in_channels}"
        e2, _ = self._p3d(x, b, T, P["encoder2"][0])
        e2, e2_32 = self._p3d(e2, b, T, P["encoder2"][1], want="both" if P["encoder4"][0][2][3] else "s")     # also the residual of encoder4.0
        c_e2pre = self._condense(e2, b, T, P["condense2"])
        e4, _ = self._p3d(e2, b, T, P["encoder4"][0], x_f32=e2_32)
        e4, e4_32 = self._p3d(e4, b, T, P["encoder4"][1], want="both" if P["res"] else "s")
        c_e4pre = self._condense(e4, b, T, P["condense4_pre"])
        for i, blk in enumerate(P["res"]):
            e4, e4_32 = self._p3d(e4, b, T, blk, want="both" if i + 1 < len(P["res"]) else "s", x_f32=e4_32)
        y = self._condense(e4, b, T, P["condense4_post"])
        for pk, d in zip(P["middle"], (8, 4, 2, 1)):
            y, _ = self._cv(y, pk, "s", pad=d, dil=d)
        y, _ = self._cv(y, P["decoder2"][0], "s", x1=c_e4pre, pad=1, upsample=True)        # cat(filled, pre) + nearest x2
        y, _ = self._cv(y, P["decoder2"][1], "s", pad=1)
        y, _ = self._cv(y, P["decoder2"][2], "s", pad=1)
        y, _ = self._cv(y, P["decoder"][0], "s", x1=c_e2pre, pad=1, upsample=True)
        y, _ = self._cv(y, P["decoder"][1], "f32", pad=1)                                  # feeds the Cout = 2 VALU conv: fp32
        fbuf = torch.zeros(b, H, W, 4, dtype=torch.float32, device=dev)               # flow in channels 0..1, zero pad for the edge head
        self._block(y, P["decoder"][2], act=None, pad=1, out=fbuf[..., :2])
        flow = ops.nhwc_to_nchw(fbuf[..., :2])
        E = P["edge"]
        pr, pr32 = self._cv(fbuf, E[0], "both", pad=1)
        ed, _ = self._cv(pr, E[1], "s", pad=1)
        ed, _ = self._cv(ed, E[2], "f32", act=None, pad=1, slope=0.01, epi="add", aux1=pr32, act2="lrelu")   # LeakyReLU() default 0.01
        if E[3][1] is None:
            edge = ops.conv2d(ed, E[3][0], pad=0, act="sigmoid", out_nchw=True)
        else:
            eo = self._block(ed, E[3], act=None, pad=0)
            edge = ops.nhwc_to_nchw(ops.axpby(eo.reshape(-1, 1), act="sigmoid").view(eo.shape))
        return flow, edge


class Model(nn.Module):
    """Drop-in for LAFC.models.lafc.Model (LAFC/models/lafc.py:6-15)."""

    def __init__(self, config):
        super().__init__()
        self.net = P3DNet(config['num_flows'], config['cnum'], config['in_channel'], config['PASSMASK'], config['use_residual'],
                          config['resBlocks'], config['use_bias'], config['conv_type'], config['init_weights'])

    def forward(self, flows, masks, edges=None):
        return self.net(flows, masks, edges)


DEFAULT_CONFIG = dict(num_flows=3, cnum=48, in_channel=3, PASSMASK=1, use_residual=1, resBlocks=1, use_bias=1,
                      conv_type='vanilla', init_weights=1)   # LAFC/config/train.yaml:51-65
